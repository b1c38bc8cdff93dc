"""TEST INFRASTRUCTURE / CPU BASELINE ONLY -- ctypes wrapper of the C++/OpenMP twin (hdg_cpu.cpp).

Loaded only by tests/, __graft_entry__ and bench.py's cpu_baseline leg.  The product never imports this module
(tests/test_host.py checks that) and has no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhdg_cpu.so")
SRC = os.path.join(_HERE, "hdg_cpu.cpp")
_DEPS = [SRC, os.path.join(_HERE, "..", "..", "include", "hdg_mi355x.h"),
         os.path.join(_HERE, "..", "..", "incompressibleeulerhdg_amd", "csrc", "hdg_tables.hpp")]
_lib = None


def build(force=False, verbose=False):
    """g++ -O3 -march=native -fopenmp; -march=native is resolved on the machine that builds, so the GPU box rebuilds
    when its CPU differs (the check below)."""
    stamp = LIB_PATH + ".host"
    host = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0] if os.path.exists("/proc/cpuinfo") else ""
    fresh = (os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in _DEPS)
             and os.path.exists(stamp) and open(stamp).read() == host)
    if fresh and not force:
        return LIB_PATH
    cmd = ["g++", "-O3", "-march=native", "-fopenmp", "-std=c++17", "-shared", "-fPIC", "-o", LIB_PATH, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    open(stamp, "w").write(host)
    return LIB_PATH


def host_threads():
    """Threads the twin may use: HDG_CPU_THREADS, else the CPUs this process is allowed to run on (affinity mask and
    cgroup quota), capped at 16 -- the CPU share of a one-GPU box.  OpenMP's own default (every core it can see) would
    oversubscribe a container that is limited to a share of the machine."""
    if os.environ.get("HDG_CPU_THREADS"):
        return max(1, int(os.environ["HDG_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.hdgcpu_last_error.restype = C.c_char_p
        _lib.hdgcpu_last_error.argtypes = [C.c_void_p]
        _lib.hdgcpu_set_num_threads(C.c_int(host_threads()))
    return _lib


_dp = C.POINTER(C.c_double)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def stream_triad_gbs(n=1 << 26, reps=5):
    """GB/s of an OpenMP stream triad on the twin's threads: the host memory bandwidth beside the CPU baseline."""
    lib = load()
    lib.hdgcpu_stream_triad_gbs.restype = C.c_double
    lib.hdgcpu_stream_triad_gbs.argtypes = [C.c_long, C.c_int]
    return float(lib.hdgcpu_stream_triad_gbs(n, reps))


class CpuTwin:
    """Same keyword arguments as incompressibleeulerhdg_amd._lib.Engine (the hdg_config fields)."""

    def __init__(self, **kw):
        from incompressibleeulerhdg_amd._lib import HDG_MAX_STAGES, hdg_config  # the struct layout of the C-ABI header

        self.lib = load()
        cfg = hdg_config()
        s = int(kw["nstages"])
        cfg.nx = cfg.ny = int(kw["nx"])
        cfg.degree, cfg.dt = int(kw["degree"]), float(kw["dt"])
        cfg.flux_upwind = 1 if kw.get("flux", "upwind") == "upwind" else 0
        cfg.use_projection, cfg.n_richardson = 1, int(kw.get("n_richardson", 2))
        cfg.tau, cfg.alpha_penalty, cfg.nstages = 1.0, 1.0, s
        for name in ("a_expl", "a_impl"):
            m = np.asarray(kw[name], dtype=float).reshape(-1)
            for i in range(s * s):
                getattr(cfg, name)[i] = m[i]
        for name in ("b_expl", "b_impl", "c_expl"):
            v = np.asarray(kw[name], dtype=float).reshape(-1)
            for i in range(min(len(v), HDG_MAX_STAGES + (1 if name == "b_impl" else 0))):
                getattr(cfg, name)[i] = v[i]
        cfg.equispaced_nodes = 0
        cfg.tent_rtol, cfg.tent_maxit, cfg.gmres_restart = 1e-10, 2000, int(kw.get("gmres_restart", 8))
        cfg.trace_rtol, cfg.trace_maxit = 1e-12, 10000
        self.h = C.c_void_p()
        rc = self.lib.hdgcpu_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(self.lib.hdgcpu_last_error(None).decode())
        nc, ne, nu, np_, nl = C.c_long(), C.c_long(), C.c_int(), C.c_int(), C.c_int()
        self.lib.hdgcpu_get_sizes(self.h, C.byref(nc), C.byref(ne), C.byref(nu), C.byref(np_), C.byref(nl))
        self.n_cells, self.n_edges, self.n_u, self.n_p, self.n_l = nc.value, ne.value, nu.value, np_.value, nl.value
        self.shape_Q, self.shape_p, self.shape_l = (nc.value * nu.value, 2), (nc.value * np_.value,), (ne.value * nl.value,)
        self.nstages = s
        self.n_total = nc.value * (2 * nu.value + np_.value) + ne.value * nl.value
        self.threads = self.lib.hdgcpu_num_threads()

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.hdgcpu_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.hdgcpu_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @staticmethod
    def _a(x, shape):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == tuple(shape), (x.shape, shape)
        return x

    def set_state(self, Q, p):
        Q, p = self._a(Q, self.shape_Q), self._a(p, self.shape_p)
        self._ck(self.lib.hdgcpu_set_state(self.h, _p(Q), _p(p)))

    def get_state(self):
        Q, p, l = np.empty(self.shape_Q), np.empty(self.shape_p), np.empty(self.shape_l)
        self._ck(self.lib.hdgcpu_get_state(self.h, _p(Q), _p(p), _p(l)))
        return Q, p, l

    def set_forcing_profile(self, f):
        f = self._a(f, self.shape_Q)
        self._ck(self.lib.hdgcpu_set_forcing_profile(self.h, _p(f)))

    def set_forcing_nodal(self, slot, f):
        f = self._a(f, self.shape_Q)
        self._ck(self.lib.hdgcpu_set_forcing_nodal(self.h, C.c_int(slot), _p(f)))

    def set_forcing_scale(self, slot, scale):
        self._ck(self.lib.hdgcpu_set_forcing_scale(self.h, C.c_int(slot), C.c_double(scale)))

    def reconstruct_trace(self):
        self._ck(self.lib.hdgcpu_reconstruct_trace(self.h))

    def step(self):
        self._ck(self.lib.hdgcpu_step(self.h))

    def run_separable(self, scales):
        scales = np.ascontiguousarray(scales, dtype=np.float64)
        self._ck(self.lib.hdgcpu_run_separable(self.h, C.c_int(scales.shape[0]), _p(scales)))

    def iteration_stats(self, reset=False):
        sums, cnt = np.zeros(4), np.zeros(4, dtype=np.int64)
        self._ck(self.lib.hdgcpu_get_iteration_stats(self.h, _p(sums), cnt.ctypes.data_as(C.POINTER(C.c_long)), C.c_int(1 if reset else 0)))
        return sums, cnt

    TIMER_LABELS = ("timestep", "bdm_projection", "tentative_velocity_solve", "pressure_solve")

    def timers(self, reset=False):
        """wall-clock per PerformanceLog label of the reference (logging.py:34-60): {label: (seconds, calls)}"""
        sec, cnt = np.zeros(4), np.zeros(4, dtype=np.int64)
        self._ck(self.lib.hdgcpu_get_timers(self.h, _p(sec), cnt.ctypes.data_as(C.POINTER(C.c_long)), C.c_int(1 if reset else 0)))
        return {lab: (float(sec[k]), int(cnt[k])) for k, lab in enumerate(self.TIMER_LABELS)}

    def node_coordinates(self):
        xq, xp = np.empty(self.shape_Q), np.empty((self.n_cells * self.n_p, 2))
        self._ck(self.lib.hdgcpu_node_coordinates(self.h, _p(xq), _p(xp)))
        return xq, xp

    def project_bdm_nodal(self, Q):
        Q, out = self._a(Q, self.shape_Q), np.empty(self.shape_Q)
        self._ck(self.lib.hdgcpu_project_bdm_nodal(self.h, _p(Q), _p(out)))
        return out

    def apply_advection(self, Qstar, x, gamma):
        Qstar, x, y = self._a(Qstar, self.shape_Q), self._a(x, self.shape_Q), np.empty(self.shape_Q)
        self._ck(self.lib.hdgcpu_apply_advection(self.h, _p(Qstar), _p(x), C.c_double(gamma), _p(y)))
        return y

    def apply_trace_operator(self, lam):
        lam, out = self._a(lam, self.shape_l), np.empty(self.shape_l)
        self._ck(self.lib.hdgcpu_apply_trace_operator(self.h, _p(lam), _p(out)))
        return out

    def apply_weak_divergence(self, Q):
        Q, out = self._a(Q, self.shape_Q), np.empty(self.shape_p)
        self._ck(self.lib.hdgcpu_apply_weak_divergence(self.h, _p(Q), _p(out)))
        return out
