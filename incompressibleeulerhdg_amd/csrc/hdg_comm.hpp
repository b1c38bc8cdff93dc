// Inter-rank transport of the strip-partitioned engine (one process per GPU).
//
// The path needs exactly three exchange patterns:
//   * neighbour halo rows (trace rows / velocity rows across a partition cut)      -> exchange()
//   * a handful of scalars per Krylov iteration (dots, norms, pressure mean)       -> allreduce_sum()
//   * the P1 coarse-grid residual of the trace preconditioner (replicated solve)   -> allgather()
//
// Backends:
//   CommSelf  single rank, everything is a no-op.
//   CommRccl  RCCL on the engine's stream: ncclSend/ncclRecv pairs over the direct xGMI link to
//             the (at most two) strip neighbours, ncclAllReduce / ncclAllGather for the rest.
//             Nothing synchronises with the host.
//   CommShm   host-staged exchange through a POSIX shared-memory segment (single node).  Exists
//             so that the partition / halo / reduction logic can be exercised by several processes
//             that share ONE GPU (RCCL refuses duplicate devices), and as a fallback.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace hdg {

struct CommError {
  std::string msg;
};

struct Comm {
  int rank = 0, size = 1;
  bool failed = false;  // a collective of this rank failed: tear the communicator down without waiting for the peers
  virtual ~Comm() {}
  // send `n` doubles from slo to rank-1 and from shi to rank+1; receive the neighbours' messages
  // into rlo (from rank-1) and rhi (from rank+1).  Missing neighbours are skipped.
  virtual void exchange(const double* slo, double* rlo, const double* shi, double* rhi, size_t n, hipStream_t st) {}
  virtual void allreduce_sum(double* dev, int n, hipStream_t st) {}
  // every rank contributes n doubles; recv holds size*n doubles in rank order
  virtual void allgather(const double* send, double* recv, size_t n, hipStream_t st) {
    if (send != recv) (void)hipMemcpyAsync(recv, send, n * sizeof(double), hipMemcpyDeviceToDevice, st);
  }
  virtual const char* name() const { return "self"; }
  // number of ranks the transport itself reports (RCCL: ncclCommCount), for the bench line
  virtual int transport_size() const { return size; }
  // deferred errors of earlier asynchronous operations (RCCL: ncclCommGetAsyncError); throws CommError
  virtual void check_async() {}
};

// ------------------------------------------------------------------------------------------------
struct CommRccl : Comm {
  ncclComm_t comm = nullptr;
  CommRccl(int rank_, int size_, const char* id128) {
    rank = rank_;
    size = size_;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclResult_t r = ncclCommInitRank(&comm, size, id, rank);
    if (r != ncclSuccess) throw CommError{std::string("ncclCommInitRank: ") + ncclGetErrorString(r)};
    // the communicator must be the one this rank believes it is part of: a mismatch (stale token, wrong WORLD_SIZE) would
    // otherwise surface as a hang in the first send / recv
    int cnt = -1, me = -1;
    if (ncclCommCount(comm, &cnt) != ncclSuccess || ncclCommUserRank(comm, &me) != ncclSuccess || cnt != size || me != rank) {
      ncclCommAbort(comm);
      comm = nullptr;
      throw CommError{"RCCL communicator does not match the launch: size " + std::to_string(cnt) + " (expected " +
                      std::to_string(size) + "), rank " + std::to_string(me) + " (expected " + std::to_string(rank) + ")"};
    }
  }
  int transport_size() const override {
    int cnt = -1;
    return (comm && ncclCommCount(comm, &cnt) == ncclSuccess) ? cnt : -1;
  }
  void check_async() override {
    if (!comm) return;
    ncclResult_t st = ncclSuccess;
    ncclResult_t r = ncclCommGetAsyncError(comm, &st);
    if (r != ncclSuccess || (st != ncclSuccess && st != ncclInProgress)) {
      failed = true;
      throw CommError{std::string("RCCL asynchronous error: ") + ncclGetErrorString(r != ncclSuccess ? r : st)};
    }
  }
  ~CommRccl() override {
    // after a rank-local failure the peers may be blocked inside a collective: abort (does not wait for them)
    if (comm) { if (failed) ncclCommAbort(comm); else ncclCommDestroy(comm); }
  }
  void ck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) { failed = true; throw CommError{std::string(what) + ": " + ncclGetErrorString(r)}; }
  }
  void exchange(const double* slo, double* rlo, const double* shi, double* rhi, size_t n, hipStream_t st) override {
    ck(ncclGroupStart(), "ncclGroupStart");
    if (rank > 0) {
      ck(ncclSend(slo, n, ncclDouble, rank - 1, comm, st), "ncclSend");
      ck(ncclRecv(rlo, n, ncclDouble, rank - 1, comm, st), "ncclRecv");
    }
    if (rank < size - 1) {
      ck(ncclSend(shi, n, ncclDouble, rank + 1, comm, st), "ncclSend");
      ck(ncclRecv(rhi, n, ncclDouble, rank + 1, comm, st), "ncclRecv");
    }
    ck(ncclGroupEnd(), "ncclGroupEnd");
  }
  // self-test only (hdg_rccl_selftest): the grouped send / recv pattern of exchange() with THIS rank as both neighbours
  // (RCCL matches sends and receives between the same pair of ranks in call order: slo -> rlo, shi -> rhi)
  void exchange_loopback(const double* slo, double* rlo, const double* shi, double* rhi, size_t n, hipStream_t st) {
    ck(ncclGroupStart(), "ncclGroupStart");
    ck(ncclSend(slo, n, ncclDouble, rank, comm, st), "ncclSend");
    ck(ncclRecv(rlo, n, ncclDouble, rank, comm, st), "ncclRecv");
    ck(ncclSend(shi, n, ncclDouble, rank, comm, st), "ncclSend");
    ck(ncclRecv(rhi, n, ncclDouble, rank, comm, st), "ncclRecv");
    ck(ncclGroupEnd(), "ncclGroupEnd");
  }
  void allreduce_sum(double* dev, int n, hipStream_t st) override {
    ck(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, comm, st), "ncclAllReduce");
  }
  void allgather(const double* send, double* recv, size_t n, hipStream_t st) override {
    ck(ncclAllGather(send, recv, n, ncclDouble, comm, st), "ncclAllGather");
  }
  const char* name() const override { return "rccl"; }
};

// ------------------------------------------------------------------------------------------------
// Shared-memory backend.  Segment layout:
//   Header[size]            per rank: epochs of the three collective kinds (cache-line padded)
//   halo slots              per rank: from_lower[cap_halo], from_upper[cap_halo]
//   reduce slots            per rank: 2 x double[64]            (double buffered by epoch parity)
//   gather slots            per rank: double[cap_gather]
// All ranks issue the same sequence of collectives, so one epoch counter per kind orders them.
struct CommShm : Comm {
  struct alignas(64) Header {
    std::atomic<uint64_t> halo_sent, halo_read, red_sent, gat_sent, gat_read, attached;
  };
  std::string shm_name;
  size_t cap_halo, cap_gather, total = 0;
  char* base = nullptr;
  uint64_t e_halo = 0, e_red = 0, e_gat = 0;
  std::vector<double> stage;

  Header* hdr(int r) const { return reinterpret_cast<Header*>(base) + r; }
  double* halo_slot(int r, int which) const {  // which: 0 = from lower, 1 = from upper
    return reinterpret_cast<double*>(base + sizeof(Header) * size) + ((size_t)r * 2 + which) * cap_halo;
  }
  double* red_slot(int r, int parity) const {
    return reinterpret_cast<double*>(base + sizeof(Header) * size) + (size_t)size * 2 * cap_halo + ((size_t)r * 2 + parity) * 64;
  }
  double* gat_slot(int r) const {
    return reinterpret_cast<double*>(base + sizeof(Header) * size) + (size_t)size * 2 * cap_halo + (size_t)size * 2 * 64 +
           (size_t)r * cap_gather;
  }
  static void wait_ge(const std::atomic<uint64_t>& a, uint64_t v, const char* what) {
    auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (a.load(std::memory_order_acquire) < v) {
      if (++spins > 200) {
        std::this_thread::yield();
        if ((spins & 1023) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0)
          throw CommError{std::string("shared-memory transport timed out waiting for ") + what};
      }
    }
  }
  CommShm(int rank_, int size_, const char* name_, size_t cap_halo_, size_t cap_gather_)
      : shm_name(name_), cap_halo(cap_halo_), cap_gather(cap_gather_) {
    rank = rank_;
    size = size_;
    total = sizeof(Header) * size + sizeof(double) * ((size_t)size * 2 * cap_halo + (size_t)size * 2 * 64 + (size_t)size * cap_gather);
    int fd = shm_open(shm_name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) throw CommError{"shm_open failed for " + shm_name};
    if (ftruncate(fd, (off_t)total) != 0) { close(fd); throw CommError{"ftruncate failed for " + shm_name}; }
    base = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { base = nullptr; throw CommError{"mmap failed for " + shm_name}; }
    // a fresh segment is zero filled by the kernel; announce this rank and wait for the others
    hdr(rank)->attached.store(1, std::memory_order_release);
    for (int r = 0; r < size; r++) wait_ge(hdr(r)->attached, 1, "peers to attach");
    // every rank holds its mapping now: the name can go (a crashed run then leaves nothing behind in /dev/shm)
    if (rank == 0) shm_unlink(shm_name.c_str());
    stage.resize(std::max<size_t>(std::max(cap_halo * 2, cap_gather), 64));
  }
  ~CommShm() override {
    if (base) {
      munmap(base, total);
    }
  }
  void exchange(const double* slo, double* rlo, const double* shi, double* rhi, size_t n, hipStream_t st) override {
    if (n > cap_halo) throw CommError{"halo message exceeds the shared-memory slot"};
    const uint64_t e = ++e_halo;
    const bool lo = rank > 0, hi = rank < size - 1;
    if (lo) (void)hipMemcpyAsync(stage.data(), slo, n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (hi) (void)hipMemcpyAsync(stage.data() + cap_halo, shi, n * sizeof(double), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    // the receiver must have consumed the previous message before its slot is overwritten
    if (lo) { wait_ge(hdr(rank - 1)->halo_read, e - 1, "lower neighbour (read)"); std::memcpy(halo_slot(rank - 1, 1), stage.data(), n * sizeof(double)); }
    if (hi) { wait_ge(hdr(rank + 1)->halo_read, e - 1, "upper neighbour (read)"); std::memcpy(halo_slot(rank + 1, 0), stage.data() + cap_halo, n * sizeof(double)); }
    hdr(rank)->halo_sent.store(e, std::memory_order_release);
    if (lo) { wait_ge(hdr(rank - 1)->halo_sent, e, "lower neighbour (send)"); (void)hipMemcpyAsync(rlo, halo_slot(rank, 0), n * sizeof(double), hipMemcpyHostToDevice, st); }
    if (hi) { wait_ge(hdr(rank + 1)->halo_sent, e, "upper neighbour (send)"); (void)hipMemcpyAsync(rhi, halo_slot(rank, 1), n * sizeof(double), hipMemcpyHostToDevice, st); }
    (void)hipStreamSynchronize(st);
    hdr(rank)->halo_read.store(e, std::memory_order_release);
  }
  void allreduce_sum(double* dev, int n, hipStream_t st) override {
    if (n > 64) throw CommError{"allreduce of more than 64 scalars"};
    const uint64_t e = ++e_red;
    double loc[64];
    (void)hipMemcpyAsync(loc, dev, n * sizeof(double), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    // slots are double buffered by epoch parity: parity e&1 was last read in epoch e-2, and a rank
    // publishes red_sent = e-1 only after it finished those reads -> wait for every rank's e-1.
    if (e > 1) for (int r = 0; r < size; r++) wait_ge(hdr(r)->red_sent, e - 1, "allreduce (previous epoch)");
    std::memcpy(red_slot(rank, (int)(e & 1)), loc, n * sizeof(double));
    hdr(rank)->red_sent.store(e, std::memory_order_release);
    double acc[64];
    for (int k = 0; k < n; k++) acc[k] = 0.0;
    for (int r = 0; r < size; r++) {  // fixed rank order: bitwise identical result on every rank
      wait_ge(hdr(r)->red_sent, e, "allreduce");
      const double* sl = red_slot(r, (int)(e & 1));
      for (int k = 0; k < n; k++) acc[k] += sl[k];
    }
    (void)hipMemcpyAsync(dev, acc, n * sizeof(double), hipMemcpyHostToDevice, st);
    (void)hipStreamSynchronize(st);
  }
  void allgather(const double* send, double* recv, size_t n, hipStream_t st) override {
    if (n > cap_gather) throw CommError{"allgather message exceeds the shared-memory slot"};
    const uint64_t e = ++e_gat;
    (void)hipMemcpyAsync(stage.data(), send, n * sizeof(double), hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    for (int r = 0; r < size; r++) wait_ge(hdr(r)->gat_read, e - 1, "allgather (previous epoch)");
    std::memcpy(gat_slot(rank), stage.data(), n * sizeof(double));
    hdr(rank)->gat_sent.store(e, std::memory_order_release);
    for (int r = 0; r < size; r++) {
      wait_ge(hdr(r)->gat_sent, e, "allgather");
      (void)hipMemcpyAsync(recv + (size_t)r * n, gat_slot(r), n * sizeof(double), hipMemcpyHostToDevice, st);
    }
    (void)hipStreamSynchronize(st);
    hdr(rank)->gat_read.store(e, std::memory_order_release);
  }
  const char* name() const override { return "shm"; }
};

}  // namespace hdg
