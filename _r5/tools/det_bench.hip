// Development harness (not part of the product): the determinizer's device code on raw lattices read from files, one workgroup
// per lattice, against the HOST build of the same header -- times per lattice, phase timers, correctness (arc multiset).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I asr-decoder_amd/csrc -o tools/det_bench tools/det_bench.hip
//   tools/det_bench [--variant N] [--reps R] lat0.bin lat1.bin ...
// File: int32 S, A; int32 is_final[S]; A x {int32 src, dst, ilabel (transition-id), olabel (word); float graph, acoustic}
// (tools/det_bench_data.py writes them from the reference's on-disk lattices).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "wfst_determinize_wave.h"   // (includes wfst_determinize.h with the many-lane Successor on the device)

using namespace wfst;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Lat {
  int S = 0, A = 0;
  std::vector<int32_t> fin, off;
  std::vector<DetArc> arcs;
};

static bool read_lat(const char *path, Lat &L) {
  FILE *f = fopen(path, "rb");
  if (!f) return false;
  int32_t hdr[2];
  if (fread(hdr, 4, 2, f) != 2) { fclose(f); return false; }
  L.S = hdr[0]; L.A = hdr[1];
  L.fin.resize(L.S);
  if (fread(L.fin.data(), 4, L.S, f) != (size_t)L.S) { fclose(f); return false; }
  struct Rec { int32_t src, dst, il, ol; float g, ac; };
  std::vector<Rec> r(L.A);
  if (fread(r.data(), sizeof(Rec), L.A, f) != (size_t)L.A) { fclose(f); return false; }
  fclose(f);
  L.off.assign((size_t)L.S + 1, 0);
  for (auto &x : r) L.off[(size_t)x.src + 1]++;
  for (int s = 0; s < L.S; ++s) L.off[(size_t)s + 1] += L.off[s];
  L.arcs.resize(L.A);
  std::vector<int32_t> cur(L.off.begin(), L.off.end() - 1);
  for (auto &x : r) {
    DetArc d; d.ilabel = x.ol; d.olabel = x.il; d.w1 = x.g; d.w2 = x.ac; d.to = x.dst;
    L.arcs[(size_t)cur[x.src]++] = d;
  }
  for (int s = 0; s < L.S; ++s)
    std::stable_sort(L.arcs.begin() + L.off[s], L.arcs.begin() + L.off[(size_t)s + 1], [](const DetArc &x, const DetArc &y) { return x.ilabel < y.ilabel; });
  return true;
}

struct OutArc { int32_t src, ilabel, next; uint32_t w1, w2; };
static bool arc_lt(const OutArc &a, const OutArc &b) { return memcmp(&a, &b, sizeof(OutArc)) < 0; }
static bool operator==(const OutArc &a, const OutArc &b) { return memcmp(&a, &b, sizeof(OutArc)) == 0; }

// canonical form of a determinized lattice whatever its state numbering: arcs as (ilabel, w1, w2, is-final) sorted -- the
// numbering-independent multiset the tests compare (tests/test_gpu_determinize.py)
static std::vector<OutArc> canon(const DetOutArc *a, int n) {
  std::vector<OutArc> v(n);
  for (int i = 0; i < n; ++i) {
    v[i].src = 0; v[i].ilabel = a[i].ilabel; v[i].next = a[i].next < 0 ? -1 : 0;
    memcpy(&v[i].w1, &a[i].w1, 4); memcpy(&v[i].w2, &a[i].w2, 4);
  }
  std::sort(v.begin(), v.end(), arc_lt);
  return v;
}

constexpr int kThreads = 256;
constexpr int kLowTmp = 1024;

struct Job {
  const int32_t *off; const DetArc *arcs; const int32_t *fin;
  int32_t S, A;
  int32_t *ws;            // det_words(caps, S)
  DetOutArc *out; int32_t *res;   // res {os_n, oa_n, err, tr_n, timers...}
  long long *timers;      // [16]
};

__global__ __launch_bounds__(kThreads) void det_kernel_lane(const Job *jobs, DetCaps caps) {   // the product's kernel as of round 4
  const Job J = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  __shared__ DetWs W;
  __shared__ DetElem s_tb[kLowTmp], s_tc[kLowTmp];
  __shared__ int s_err;
  if (tid == 0) {
    W.n_states = J.S; W.n_arcs = J.A; W.off = J.off; W.arcs = J.arcs; W.is_final = J.fin; W.delta = 1.0f / 1024;
    det_carve(W, J.ws, caps, J.S);
    W.tb_lo = s_tb; W.tc_lo = s_tc; W.tmp_lo = kLowTmp;
  }
  __syncthreads();
  const int32_t hcap_full = W.tr_hcap;
  long long t0 = clock64();
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) {
      int32_t h = hcap_full;
      if (attempt == 0) { h = 4096; while (h < 16 * J.S && h < hcap_full) h <<= 1; }
      W.tr_hcap = h < hcap_full ? h : hcap_full;
    }
    __syncthreads();
    det_init(W, tid, kThreads);
    __syncthreads();
    if (tid == 0) s_err = det_run(W);
    __syncthreads();
    if (!(s_err == 1 && W.tr_hcap < hcap_full)) break;
  }
  if (tid == 0) {
    J.res[0] = W.os_n; J.res[1] = W.oa_n; J.res[2] = s_err; J.res[3] = W.tr_n;
    J.timers[0] = clock64() - t0;
    for (int i = 0; i < W.oa_n; ++i) J.out[i] = W.oarcs[i];
  }
}

__global__ __launch_bounds__(kThreads) void det_kernel_lane_private(const Job *jobs, DetCaps caps) {   // the same, W in registers
  const Job J = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  __shared__ DetWs Ws;
  __shared__ DetElem s_tb[kLowTmp], s_tc[kLowTmp];
  if (tid == 0) {
    Ws.n_states = J.S; Ws.n_arcs = J.A; Ws.off = J.off; Ws.arcs = J.arcs; Ws.is_final = J.fin; Ws.delta = 1.0f / 1024;
    det_carve(Ws, J.ws, caps, J.S);
    Ws.tb_lo = s_tb; Ws.tc_lo = s_tc; Ws.tmp_lo = kLowTmp;
    int32_t h = 4096; while (h < 16 * J.S && h < Ws.tr_hcap) h <<= 1;
    Ws.tr_hcap = h < Ws.tr_hcap ? h : Ws.tr_hcap;
  }
  __syncthreads();
  DetWs W = Ws;   // private: pointers and counters in registers
  long long t0 = clock64();
  det_init(W, tid, kThreads);
  __syncthreads();
  if (tid == 0) {
    const int err = det_run(W);
    J.res[0] = W.os_n; J.res[1] = W.oa_n; J.res[2] = err; J.res[3] = W.tr_n;
    J.timers[0] = clock64() - t0;
    for (int i = 0; i < W.oa_n; ++i) J.out[i] = W.oarcs[i];
  }
}

__global__ __launch_bounds__(kThreads) void det_kernel_wave(const Job *jobs, DetCaps caps, int variant) {
  const Job J = jobs[blockIdx.x];
  detw_run_block(J.off, J.arcs, J.fin, J.S, J.A, J.ws, caps, J.out, J.res, J.timers, variant);
}

static void launch(const Job *jobs, int n, const DetCaps &caps, int variant) {
  if (variant == 0) hipLaunchKernelGGL(det_kernel_lane, dim3((unsigned)n), dim3(kThreads), 0, 0, jobs, caps);
  else if (variant == 2) hipLaunchKernelGGL(det_kernel_lane_private, dim3((unsigned)n), dim3(kThreads), 0, 0, jobs, caps);
  else hipLaunchKernelGGL(det_kernel_wave, dim3((unsigned)n), dim3(kThreads), 0, 0, jobs, caps, variant);
}

int main(int argc, char **argv) {
  int variant = 0, reps = 3;
  std::vector<std::string> files;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--variant")) variant = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else files.push_back(argv[i]);
  }
  std::vector<Lat> lats(files.size());
  for (size_t i = 0; i < files.size(); ++i)
    if (!read_lat(files[i].c_str(), lats[i])) { fprintf(stderr, "cannot read %s\n", files[i].c_str()); return 1; }
  int maxS = 0, maxA = 0;
  for (auto &L : lats) { maxS = std::max(maxS, L.S); maxA = std::max(maxA, L.A); }
  DetCaps caps;
  const int32_t base = std::max(4096, 65536);
  caps.trie = 8 * base; caps.pool = 16 * base; caps.states = 2 * base; caps.initials = 2 * base; caps.arcs = 4 * base;
  caps.tmp = std::max(8192, 2 * (maxA + maxS));
  // ---- host reference (the same header, sequential) ----------------------------------------------------
  std::vector<std::vector<OutArc>> want(lats.size());
  std::vector<int> want_states(lats.size());
  for (size_t i = 0; i < lats.size(); ++i) {
    Lat &L = lats[i];
    std::vector<int32_t> ws((size_t)det_words(caps, L.S));
    DetWs W;
    memset(&W, 0, sizeof(W));
    W.n_states = L.S; W.n_arcs = L.A; W.off = L.off.data(); W.arcs = L.arcs.data(); W.is_final = L.fin.data(); W.delta = 1.0f / 1024;
    det_carve(W, ws.data(), caps, L.S);
    det_init(W, 0, 1);
    const int err = det_run(W);
    if (err) { fprintf(stderr, "host determinizer: err %d on %s\n", err, files[i].c_str()); return 1; }
    want[i] = canon(W.oarcs, W.oa_n);
    want_states[i] = W.os_n;
  }
  // ---- device ----------------------------------------------------------------------------------------
  std::vector<Job> jobs(lats.size());
  for (size_t i = 0; i < lats.size(); ++i) {
    Lat &L = lats[i];
    Job &J = jobs[i];
    int32_t *off, *fin, *ws, *res; DetArc *arcs; DetOutArc *out; long long *tm;
    CK(hipMalloc(&off, 4 * (L.S + 1))); CK(hipMalloc(&fin, 4 * std::max(L.S, 1))); CK(hipMalloc(&arcs, sizeof(DetArc) * std::max(L.A, 1)));
    CK(hipMemcpy(off, L.off.data(), 4 * (L.S + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(fin, L.fin.data(), 4 * L.S, hipMemcpyHostToDevice));
    CK(hipMemcpy(arcs, L.arcs.data(), sizeof(DetArc) * L.A, hipMemcpyHostToDevice));
    CK(hipMalloc(&ws, 4 * (size_t)det_words(caps, L.S) + 4 * (size_t)detw_extra_words(caps, L.S)));
    CK(hipMalloc(&out, sizeof(DetOutArc) * caps.arcs)); CK(hipMalloc(&res, 64)); CK(hipMalloc(&tm, 16 * 8));
    CK(hipMemset(tm, 0, 16 * 8));
    J.off = off; J.arcs = arcs; J.fin = fin; J.S = L.S; J.A = L.A; J.ws = ws; J.out = out; J.res = res; J.timers = tm;
  }
  Job *djobs;
  CK(hipMalloc(&djobs, sizeof(Job) * jobs.size()));
  CK(hipMemcpy(djobs, jobs.data(), sizeof(Job) * jobs.size(), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bad = 0;
  for (int r = 0; r < reps; ++r) {
    // all lattices side by side (the product's launch), then each alone
    CK(hipEventRecord(e0));
    launch(djobs, (int)jobs.size(), caps, variant);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("variant %d rep %d: %zu lattices side by side %.3f ms\n", variant, r, jobs.size(), ms);
  }
  for (size_t i = 0; i < jobs.size(); ++i) {
    CK(hipEventRecord(e0));
    launch(djobs + i, 1, caps, variant);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    int32_t res[4];
    long long tm[16];
    CK(hipMemcpy(res, jobs[i].res, 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(tm, jobs[i].timers, 128, hipMemcpyDeviceToHost));
    std::vector<DetOutArc> out(std::max(res[1], 1));
    CK(hipMemcpy(out.data(), jobs[i].out, sizeof(DetOutArc) * std::max(0, std::min(res[1], caps.arcs)), hipMemcpyDeviceToHost));
    const bool ok = res[2] == 0 && res[0] == want_states[i] && canon(out.data(), res[1]) == want[i];
    bad += !ok;
    printf("  %-28s S %6d A %6d -> states %5d arcs %5d err %d trie %6d  alone %.3f ms  %s | clk(M):", files[i].c_str(), lats[i].S, lats[i].A, res[0], res[1],
           res[2], res[3], ms, ok ? "OK" : "MISMATCH");
    for (int k = 0; k < 16; ++k) printf(" %.3f", tm[k] / 1e6);
    printf("\n");
  }
  printf("%s\n", bad ? "FAILED" : "all equal to the host build");
  return bad ? 1 : 0;
}
