"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, class surface
matches the reference, host utilities behave like the reference's."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from incompressibleeulerhdg_amd import _lib

    _lib.build_library()
    lib = _lib.load_library()
    header = open(os.path.join(ROOT, "include", "hdg_mi355x.h")).read()
    declared = set(re.findall(r"\b(hdg_[a-z0-9_]+)\s*\(", header))
    declared.discard("hdg_handle")
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), name
    # the ctypes signature table covers the same set
    assert declared == set(_lib.SIGNATURES) | {"hdg_last_error"}


def test_config_struct_layout_matches_header():
    from incompressibleeulerhdg_amd import _lib

    header = open(os.path.join(ROOT, "include", "hdg_mi355x.h")).read()
    body = header[header.index("typedef struct hdg_config {"):header.index("} hdg_config;")]
    fields = re.findall(r"\b(?:int|double)\s+([^;]+);", body)
    names = []
    for f in fields:
        for part in f.split(","):
            names.append(re.match(r"\s*([a-z_]+)", part).group(1))
    assert names == [n for n, _ in _lib.hdg_config._fields_]


def test_no_gpu_gives_an_error_code_not_a_crash():
    """Without a GPU hdg_create must fail with HDG_ERR_HIP (no CPU fallback, no abort)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from incompressibleeulerhdg_amd._lib import Engine, HDGError

    with pytest.raises(HDGError) as ei:
        Engine(nx=4, degree=1, dt=0.1, nstages=2, a_expl=[[0, 0], [1, 0]], a_impl=[[0, 0], [0, 1]], b_expl=[1, 0],
               b_impl=[0, 1], c_expl=[0, 1])
    assert ei.value.code == -2


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "incompressibleeulerhdg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), os.path.join(dirpath, f)


def test_class_surface_matches_reference():
    """Constructor / method names and argument order of hdg_imex.py:29-39,258,275,505 and
    hdg_implicit.py:17-25,52."""
    from incompressibleeulerhdg_amd import timesteppers as ts

    base = ts.IncompressibleEulerHDGIMEX
    sig = list(inspect.signature(base.__init__).parameters)
    assert sig[:9] == ["self", "mesh", "degree", "dt", "flux", "use_projection_method", "n_richardson", "label", "callbacks"]
    for name in ("IncompressibleEulerHDGIMEXImplicit", "IncompressibleEulerHDGIMEXARS2_232",
                 "IncompressibleEulerHDGIMEXARS3_443", "IncompressibleEulerHDGIMEXSSP2_332",
                 "IncompressibleEulerHDGIMEXSSP3_433"):
        cls = getattr(ts, name)
        assert issubclass(cls, base)
        assert list(inspect.signature(cls.__init__).parameters)[:8] == [
            "self", "mesh", "degree", "dt", "flux", "use_projection_method", "n_richardson", "callbacks"]
    assert list(inspect.signature(base.solve).parameters)[:7] == [
        "self", "Q_initial", "p_initial", "q_initial", "f_rhs", "T_final", "warmup"]
    for m in ("pressure_solve", "tentative_velocity_solve", "project_bdm", "get_timesteps", "_shift_pressure",
              "_reconstruct_trace"):
        assert hasattr(base, m)
    imp = ts.IncompressibleEulerHDGImplicit
    assert list(inspect.signature(imp.__init__).parameters)[:7] == [
        "self", "mesh", "degree", "dt", "flux", "use_projection_method", "callbacks"]
    assert "n_richardson" in inspect.signature(imp.__init__).parameters  # driver.py:220-228 (SURVEY C-1)


def test_tableau_properties_equal_oracle_literals():
    from incompressibleeulerhdg_amd import timesteppers as ts
    from oracle.hdg_oracle import TABLEAUX

    pairs = {"imex_implicit": ts.IncompressibleEulerHDGIMEXImplicit, "imex_ars2_232": ts.IncompressibleEulerHDGIMEXARS2_232,
             "imex_ars3_443": ts.IncompressibleEulerHDGIMEXARS3_443, "imex_ssp2_332": ts.IncompressibleEulerHDGIMEXSSP2_332,
             "imex_ssp3_433": ts.IncompressibleEulerHDGIMEXSSP3_433}
    for key, cls in pairs.items():
        t = TABLEAUX[key]
        for attr, name in (("_a_expl", "a_expl"), ("_a_impl", "a_impl"), ("_b_expl", "b_expl"), ("_b_impl", "b_impl"),
                           ("_c_expl", "c_expl")):
            got = getattr(cls, attr).fget(None)
            assert np.array_equal(got, np.asarray(t[name], dtype=float)), (key, attr)
        assert cls.nstages.fget(None) == len(t["c_expl"])
        assert cls.__name__.startswith("IncompressibleEulerHDGIMEX")


def test_performance_log_and_averager():
    from incompressibleeulerhdg_amd.auxilliary.logging import PerformanceLog, log_summary
    from incompressibleeulerhdg_amd.auxilliary.utils import Averager

    PerformanceLog.reset()
    with PerformanceLog("timestep"):
        pass

    @PerformanceLog("pressure_solve")
    def f():
        return 3

    assert f() == 3 and f() == 3
    assert len(PerformanceLog.data["timestep"]) == 1 and len(PerformanceLog.data["pressure_solve"]) == 2
    log_summary()
    a = Averager()
    for x in (1, 2, 6):
        a.update(x)
    assert a.value == 3 and a.n_samples == 3


def test_get_timesteps_and_model_problem():
    from incompressibleeulerhdg_amd.mesh import FunctionSpace, UnitSquareMesh
    from incompressibleeulerhdg_amd.model_problems import TaylorGreen
    from incompressibleeulerhdg_amd.timesteppers.common import IncompressibleEuler

    class T(IncompressibleEuler):
        def solve(self, *a, **k):
            pass

    t = T(UnitSquareMesh(4, 4), 1, 0.04)
    assert t.get_timesteps(1.0, False) == 25 and t.get_timesteps(1.0, True) == 1
    with pytest.raises(AssertionError):
        t.get_timesteps(0.05, False)
    xy = np.array([[0.0, 0.0], [0.5, 0.5], [1.0, 0.25]])
    VQ = FunctionSpace(None, "DG", 2, xy, 2)
    Vp = FunctionSpace(None, "DG", 1, xy, 1)
    mp = TaylorGreen(VQ, Vp, "exponential", 0.5)
    f = mp.f_rhs()
    assert np.allclose(f(0.3), -0.5 * np.exp(-0.15) * mp._Qs)
    assert np.allclose(mp._Qs[1], [0.0, 0.0]) and abs(mp._ps[1]) < 1e-15
    Q, p = mp.solution(0.2)
    assert np.allclose(Q.dat.data, np.exp(-0.1) * mp._Qs)
    assert TaylorGreen(VQ, Vp, "constant", 0.0).f_rhs().scale(1.0) == 0.0


def _disassemble_device_code(tmp_path):
    """gfx950 ISA of the built library: llvm-objdump extracts the offload bundle next to its input, so work on a copy."""
    import shutil
    import subprocess

    from incompressibleeulerhdg_amd import _lib

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    _lib.build_library()
    so = str(tmp_path / "libhdg_mi355x.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", so], check=True, stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    co = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert len(co) == 1, co
    return subprocess.run([objdump, "-d", str(tmp_path / co[0])], check=True, stdout=subprocess.PIPE).stdout.decode()


def _vregs(tok):
    """vector registers named by an operand token (VGPRs and AGPRs): v7 -> {("v", 7)}, a[4:7] -> {("a", 4), ...}; else {}"""
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), n) for n in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def _audit_16_byte_stores(asm):
    """Checks (a) and (b) of test_isa_store_data_hazard_audit on a disassembly; returns the number of stores audited."""
    lines = [ln.split("//")[0].strip() for ln in asm.splitlines()]
    lines = [ln for ln in lines if ln and not ln.endswith(":") and not ln.startswith(("Disassembly", "/"))]
    n_stores = 0
    for idx, ln in enumerate(lines):
        if not ln.startswith("buffer_store_dwordx4"):
            continue
        n_stores += 1
        ops = [t.strip() for t in ln[len("buffer_store_dwordx4"):].split(",")]
        # operands: vdata, vaddr, srsrc, "soffset [modifiers]"
        assert len(ops) == 4, ln
        soffset = ops[3].split()[0]
        assert not soffset.startswith(("s", "m0")), f"16-byte buffer store with an SGPR soffset: {ln}"
        data = _vregs(ops[0])
        assert len(data) == 4, ln
        waited, q = 0, idx + 1
        while waited < 2 and q < len(lines):
            nxt = lines[q]
            mnem = nxt.split()[0]
            if mnem.startswith(("s_endpgm", "s_branch", "s_cbranch", "s_setpc")):
                break  # the window ends with the basic block (the audit is per fall-through path)
            if mnem == "s_nop":
                waited += int(nxt.split()[1], 0) + 1
            else:
                if mnem.startswith("v_") and not mnem.startswith("v_cmp"):
                    dst = nxt[len(mnem):].split(",")[0].strip()
                    assert not (_vregs(dst) & data), f"store data overwritten inside the hazard window:\n  {ln}\n  {nxt}"
                waited += 1
            q += 1
    return n_stores


def test_isa_audit_detects_the_hazard_patterns():
    """The audit itself: the two patterns behind the round-2 defect are rejected, the padded forms are accepted."""
    ok = """
	buffer_store_dwordx4 v[0:3], v57, s[4:7], 0 offen          // 000: E07C1000
	s_nop 1
	v_fma_f64 v[0:1], v[8:9], v[8:9], 0
	buffer_store_dwordx4 a[16:19], v10, s[56:59], 0 offen nt
	v_add_u32_e32 v25, s1, v57
	v_add_u32_e32 v24, s0, v24
	v_accvgpr_write_b32 a16, v3
"""
    assert _audit_16_byte_stores(ok) == 2
    with pytest.raises(AssertionError, match="SGPR soffset"):
        _audit_16_byte_stores("	buffer_store_dwordx4 v[0:3], v57, s[4:7], s12 offen\n")
    with pytest.raises(AssertionError, match="hazard window"):
        _audit_16_byte_stores("	buffer_store_dwordx4 v[0:3], v57, s[4:7], 0 offen\n	v_mov_b32_e32 v2, 0\n")
    with pytest.raises(AssertionError, match="hazard window"):
        _audit_16_byte_stores("	buffer_store_dwordx4 a[4:7], v57, s[4:7], 0 offen\n	s_nop 0\n	v_accvgpr_write_b32 a5, v3\n")


def test_isa_store_data_hazard_audit(tmp_path):
    """Pins the round-2 hardware finding ('store-data hazard'): on gfx950 a 16-byte buffer store whose
    data registers are overwritten by the next VALU instruction can pick up the new value; hipcc pads that hazard with wait
    states only for stores WITHOUT a scalar offset register.  Audit of the shipped code object:
      (a) no buffer_store_dwordx4 carries an SGPR soffset (VelBuf::st puts the plane offset into the vector offset);
      (b) no VALU instruction writes a store's data registers within the 2 wait states LLVM's hazard model requires on
          gfx940+ (VmemStoreHazard; s_nop N counts N + 1).
    """
    asm = _disassemble_device_code(tmp_path)
    n_stores = _audit_16_byte_stores(asm)
    assert n_stores > 100, n_stores  # every velocity-writing kernel stores 16-byte pairs


def test_every_kernel_form_has_its_own_name(tmp_path):
    """Plain / residual form of the advection operator and the lift with / without the Chebyshev epilogue are separate
    instantiations: profiles and PMC passes tell them apart by kernel name (profiles/pmc_traffic.json is per form)."""
    asm = _disassemble_device_code(tmp_path)
    syms = set(re.findall(r"<(_ZN3hdg[A-Za-z0-9_]+)>:", asm))
    for frag in ("11k_adv_applyILi2ELb1E", "11k_adv_applyILi2ELb0E", "11k_edge_liftILi2ELb0ELi2ELb1E", "11k_edge_liftILi2ELb0ELi2ELb0E",
                 "10k_adv_mfmaILi4ELb1E", "10k_adv_mfmaILi4ELb0E"):
        assert any(frag in sy for sy in syms), frag


def test_bench_watchdog_ends_an_overrunning_phase():
    """bench.py bounds every phase of a multi-rank run (transport initialisation, warm-up, timed steps): a rank stuck in a
    collective whose peer died ends itself with exit code 4 instead of hanging the job (torchrun tears the rest down)."""
    import subprocess
    import sys

    code = ("import sys, time; sys.path.insert(0, %r); import bench; w = bench.Watchdog(0); w.arm('fast phase', 30); w.disarm();"
            "w.arm('stuck collective', 1.0); time.sleep(30); sys.exit(0)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 4, (r.returncode, r.stderr.decode()[-500:])
    assert "stuck collective" in r.stderr.decode()


def test_unit_disk_mesh_and_general_mesh_output(tmp_path):
    """Product-side UnitDiskMesh (driver.py:184-185) and the VTK writer on a general triangulation (host side only)."""
    from incompressibleeulerhdg_amd.mesh import Function, FunctionSpace, TriangleMesh, UnitDiskMesh
    from incompressibleeulerhdg_amd.output import VTKFile, cell_vertex_nodes

    for level in (0, 1, 2):
        m = UnitDiskMesh(level)
        assert m.num_cells() == 8 * 4 ** level and 2.8 < m.volume < np.pi
        r = np.linalg.norm(m.vertices, axis=1)
        assert r.max() < 1.0 + 1e-14 and np.sum(np.abs(r - 1.0) < 1e-13) == 8 * 2 ** level  # boundary vertices on the unit circle
    with pytest.raises(ValueError):
        TriangleMesh([[0, 0], [1, 0], [2, 0]], [[0, 1, 2]])
    m = UnitDiskMesh(1)
    # degree-2 lattice nodes (b outer, a inner) of every cell on its own vertices
    lat = [(a / 2.0, b / 2.0) for b in range(3) for a in range(3 - b)]
    v = m.vertices[m.cells]
    X = np.stack([v[:, 0] + (v[:, 1] - v[:, 0]) * xi + (v[:, 2] - v[:, 0]) * eta for xi, eta in lat], axis=1).reshape(-1, 2)
    V = FunctionSpace(m, "DG", 2, X)
    assert np.array_equal(cell_vertex_nodes(V), [[0, 2, 5], [0, 2, 5]])
    f = Function(V, X[:, 0] + 2 * X[:, 1], "f")
    vtu = VTKFile(str(tmp_path / "disk.pvd")).write(f)
    txt = open(vtu).read()
    assert f'NumberOfCells="{m.num_cells()}"' in txt and 'Name="f"' in txt


@pytest.mark.parametrize("k,level", [(1, 2), (2, 1), (3, 1)])
def test_general_mesh_host_side_invariants(tmp_path, k, level):
    """The host side of the general-mesh path (csrc/hdg_general.hpp, hdg_amg.hpp: plain C++, compiled here with g++, no GPU):
    condensed operator symmetric with the constants as kernel, BDM projection idempotent, the constraint rows of the
    monolithic system vanish on constants, the P1 prolongation reproduces constants, every level of the smoothed-aggregation
    hierarchy keeps symmetry / kernel / partition of unity and gets smaller, the dense pseudo-inverse of the coarsest operator
    inverts on the range, the continuous space has the mesh's volume, projects constants and gives vorticity 2 for a rotation."""
    import shutil
    import subprocess

    from incompressibleeulerhdg_amd.mesh import UnitDiskMesh

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "general_host_check"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "general_host_check.cpp")
    subprocess.run([gxx, "-std=c++17", "-O1", "-o", str(exe), src], check=True)
    m = UnitDiskMesh(level)
    with open(tmp_path / "mesh.txt", "w") as f:
        f.write(f"{len(m.vertices)} {len(m.cells)}\n")
        np.savetxt(f, m.vertices, fmt="%.17g")
        np.savetxt(f, m.cells, fmt="%d")
    out = subprocess.run([str(exe), str(tmp_path / "mesh.txt"), str(k)], check=True, capture_output=True, text=True).stdout
    v = {ln.split()[0]: ln.split()[1] for ln in out.strip().splitlines()}
    assert "error" not in v, out
    f = lambda name: float(v[name])
    assert int(v["nc"]) == 8 * 4 ** level and abs(f("volume") - m.volume) < 1e-12
    for name in ("S_asym", "S_null", "Pi_idempotent", "mu_row_constant", "psi_row_constant", "P0_constants", "amg_null", "amg_asym",
                 "amg_P_constants", "coarse_pinv", "cg_M_asym", "cg_projection_rhs_constant", "cg_vorticity_rotation"):
        assert f(name) < 1e-10, (name, v[name])
    assert int(v["amg_levels"]) >= 2 and int(v["amg_monotone"]) == 1
    p = k + 1
    assert int(v["ncg"]) == int(v["nv"]) + int(v["ne"]) * (p - 1) + int(v["nc"]) * (p - 1) * (p - 2) // 2
    assert abs(f("cg_volume") - m.volume) < 1e-11


def test_committed_pmc_traffic_belongs_to_the_committed_kernel_sources():
    """profiles/pmc_traffic.json (HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes; bench.py prints them as
    `roofline.traffic`) carries the hash of every kernel / engine source it was measured on: it must be the hash of the sources
    in the tree, or the bench line would report no traffic (and the profile set under profiles/ would be of other code)."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    assert t["csrc_sha16"] == bench.csrc_sha16(), (t["tag"], "re-collect with tools/collect_profiles.sh after changing csrc/")
    tag = t["tag"]
    for name in (f"{tag}_bench_c3.json", f"{tag}_c3_kernel_stats.csv", f"{tag}_c3_pmc_fetch.csv", f"{tag}_c3_pmc_write.csv"):
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), name
    calib = t["calibration"]  # the stream triad: FETCH_SIZE counts half of the bytes on gfx950, WRITE_SIZE all of them
    assert abs(calib["measured_factor_fetch"] - 2.0) < 0.01 and abs(calib["measured_factor_write"] - 1.0) < 0.01


def test_side_rows_of_a_leg_launch_cover_every_row_once(tmp_path):
    """csrc/hdg_side_rows.hpp (plain C++, compiled here with g++): the interleaving of tile rows and side rows of a V-cycle leg
    launch that carries a share of the condensed CG's p / x update visits every tile row and every side row exactly once, for
    1..70 tile rows and 0..400 side rows (the GPU tests cover a handful of these through whole steps)."""
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "side_rows_check"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "side_rows_check.cpp")
    subprocess.run([gxx, "-std=c++17", "-O1", "-o", str(exe), src], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok 28070"), out


def test_unit_disk_mesh_numbering_independent_invariants():
    """UnitDiskMesh (driver.py:184-185) is restated from memory of Firedrake's utility mesh on both the product and the
    oracle side -- no reference fixture covers its numbering or coordinates (parity unpinned).  What any correct
    construction must satisfy, whatever its numbering: 8 * 4^L cells, Euler's formula for a disk, every boundary vertex on
    the unit circle and none of the interior ones, positively oriented non-degenerate cells, an area that tends to pi at second
    order, and the same vertex / cell counts and area from the oracle's independent construction."""
    import numpy as np

    from incompressibleeulerhdg_amd.mesh import UnitDiskMesh
    from oracle import fem

    prev_err = None
    for level in range(0, 6):
        m = UnitDiskMesh(level)
        X, C = np.asarray(m.vertices), np.asarray(m.cells)
        nc = 8 * 4**level
        assert C.shape == (nc, 3) and len(np.unique(C)) == len(X)
        # edges: interior ones are shared by two cells, boundary ones by one
        E = np.sort(np.concatenate([C[:, [0, 1]], C[:, [1, 2]], C[:, [2, 0]]]), axis=1)
        uniq, counts = np.unique(E, axis=0, return_counts=True)
        assert set(counts) <= {1, 2}
        nb = int(np.sum(counts == 1))
        assert nb == 8 * 2**level                      # boundary edges = boundary vertices of a disk
        assert len(X) - len(uniq) + nc == 1            # Euler characteristic of a disk
        bverts = np.unique(uniq[counts == 1])
        r = np.linalg.norm(X, axis=1)
        assert len(bverts) == nb and np.allclose(r[bverts], 1.0, atol=1e-14)
        interior = np.setdiff1d(np.arange(len(X)), bverts)
        assert np.all(r[interior] < 1.0 - 1e-9)
        a, b, c = X[C[:, 0]], X[C[:, 1]], X[C[:, 2]]
        area2 = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
        assert np.all(np.abs(area2) > 1e-12)
        area = 0.5 * np.sum(np.abs(area2))
        err = np.pi - area                             # inscribed polygon: below pi, second order in h
        assert err > 0
        if prev_err is not None and level >= 2:
            assert 3.0 < prev_err / err < 5.0, (level, prev_err / err)
        prev_err = err
        # the oracle's independent construction: same counts and area (the numbering may differ)
        o = fem.unit_disk_mesh(level)
        if o is not None:
            oX, oC = np.asarray(o.vertices), np.asarray(o.cells)
            assert oC.shape == C.shape and oX.shape == X.shape
            oa, ob, oc = oX[oC[:, 0]], oX[oC[:, 1]], oX[oC[:, 2]]
            oarea = 0.5 * np.sum(np.abs((ob[:, 0] - oa[:, 0]) * (oc[:, 1] - oa[:, 1]) - (ob[:, 1] - oa[:, 1]) * (oc[:, 0] - oa[:, 0])))
            assert abs(oarea - area) < 1e-12
    assert prev_err < 2e-3
