// DUET's topological map, resident on the device, batched over the episodes of a rank.
//
// The reference keeps one python dict-of-dicts graph per episode (VLN-DUET/map_nav_src/models/graph_utils.py:43-161) and rebuilds
// position features / pair distances with python double loops every step (r2r/agent.py:98-207).  Here every episode owns dense
// G x G matrices in HBM (float64 distances, int32 intermediate-node marks) plus float64 node positions; the host only assigns
// node slots (names are strings) and sends a few hundred integers per step.
//   graph_observe_kernel     graph_utils.py:107-113 + :55-70  edges to the candidates, then the relaxation round through `cur`
//   graph_pos_fts_kernel     graph_utils.py:130-153           7 position features per listed node; hop counts expand the
//                                                             intermediate-node marks lazily, exactly like FloydGraph.path
//   graph_pair_dists_kernel  r2r/agent.py:135-139,157-160     distances between the listed nodes, float32
//   gather_rows_or_zero      agent_cmt.py:286-309             imagination rows into their sub-instruction slots
// All HBM-bound byte shuffling with a few float64 operations per element: one block per episode (or per row), coalesced rows,
// no attempt to involve the matrix cores.
#include "common.h"

// float64 geometry must round exactly like the reference's python floats: no fused multiply-add contraction in this file
#pragma clang fp contract(off)

namespace {

constexpr double UNREACHED = 95959595.0;   // graph_utils.py:45
constexpr int MAXG = 256;                   // node slots per episode the hop-count stack (uint8 node ids) can address

// correctly rounded square root: hardware estimate, then one exact-residual (fma) correction
__device__ __forceinline__ double sqrt_rn(double s) {
  const double r = sqrt(s);
  if (!(r > 0.0) || isinf(r)) return r;
  return r + __builtin_fma(-r, r, s) / (2.0 * r);
}

__global__ void graph_init_kernel(double* dis, int* via, unsigned char* seen, long n_pairs, long n_nodes) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pairs) { dis[i] = UNREACHED; via[i] = -1; }
  if (i < n_nodes) seen[i] = 0;
}

// one block per episode
__global__ __launch_bounds__(256) void graph_observe_kernel(double* pos, double* dis, int* via, unsigned char* seen, const int* cur,
                                                            const int* cand, const double* cur_pos, const double* cand_pos,
                                                            const double* cand_dist, const int* n_nodes, int G, int C) {
  const int b = blockIdx.x, c = cur[b];
  if (c < 0) return;                                            // episode ended: the reference skips update_graph (agent.py:606-608)
  double* P = pos + (long)b * G * 3;
  double* D = dis + (long)b * G * G;
  int* V = via + (long)b * G * G;
  if (threadIdx.x == 0) {                                       // candidates in order: later duplicates only win if shorter (add_edge)
    const double ax = cur_pos[b * 3], ay = cur_pos[b * 3 + 1], az = cur_pos[b * 3 + 2];
    P[c * 3] = ax; P[c * 3 + 1] = ay; P[c * 3 + 2] = az;
    for (int j = 0; j < C; ++j) {
      const int u = cand[b * C + j];
      if (u < 0) continue;
      const double* q = cand_pos + ((long)b * C + j) * 3;
      P[u * 3] = q[0]; P[u * 3 + 1] = q[1]; P[u * 3 + 2] = q[2];
      // edge length: by default the exact float64 value; the host may pass the reference's own (python `dx**2` goes through libm
      // pow, which differs from dx*dx by one ulp for ~0.1 % of the values, and the map compares these lengths with `<`)
      const double dx = q[0] - ax, dy = q[1] - ay, dz = q[2] - az;
      const double d = cand_dist ? cand_dist[b * C + j] : sqrt_rn((dx * dx + dy * dy) + dz * dz);
      if (d < D[(long)c * G + u]) {
        D[(long)c * G + u] = d; D[(long)u * G + c] = d;
        V[(long)c * G + u] = -1; V[(long)u * G + c] = -1;
      }
    }
    seen[(long)b * G + c] = 1;
  }
  __syncthreads();
  // relaxation through c: row c and column c never change in this round (D[c][c] stays UNREACHED), so pairs are independent
  const int n = n_nodes[b];
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) {
    const int x = i / n, y = i - x * n;
    if (x == y) continue;
    const double t = D[(long)x * G + c] + D[(long)c * G + y];
    if (t < D[(long)x * G + y]) { D[(long)x * G + y] = t; V[(long)x * G + y] = c; }
  }
}

// grid (ceil(N / 64), B), 64 threads: one listed node per thread
__global__ __launch_bounds__(64) void graph_pos_fts_kernel(const double* pos, const double* dis, const int* via, const int* cur,
                                                           const int* nodes, const double* heading, const double* elevation,
                                                           float* out, long ld_row, long ld_batch, int* status, int G, int N, int A) {
  __shared__ unsigned char stack[64][MAXG][2];
  const int b = blockIdx.y, r = blockIdx.x * 64 + threadIdx.x;
  if (r >= N) return;
  float* o = out + (long)b * ld_batch + (long)r * ld_row;
  const int j = nodes[(long)b * N + r], c = cur[b];
  if (j == -2 || c < 0) {                                        // padding row
    for (int k = 0; k < A + 3; ++k) o[k] = 0.f;
    return;
  }
  float ang_h = 0.f, ang_e = 0.f;
  double f4 = 0, f5 = 0, f6 = 0;
  if (j >= 0) {
    const double* P = pos + (long)b * G * 3;
    const double dx = P[j * 3] - P[c * 3], dy = P[j * 3 + 1] - P[c * 3 + 1], dz = P[j * 3 + 2] - P[c * 3 + 2];
    const double sxy = dx * dx + dy * dy;
    const double xy = fmax(sqrt_rn(sxy), 1e-8), xyz = fmax(sqrt_rn(sxy + dz * dz), 1e-8);
    double h = asin(dx / xy);
    if (P[j * 3 + 1] < P[c * 3 + 1]) h = 3.141592653589793 - h;
    ang_h = (float)(h - heading[b]);                             // the reference casts the ANGLES to float32 before sin / cos
    ang_e = (float)(asin(dz / xyz) - elevation[b]);
    // hop count = len(path(c, j)): expand the marks depth-first; every pop either finishes a pair or replaces it by two
    const int* V = via + (long)b * G * G;
    int hops = 0, sp = 0, pops = 0;
    bool bad = false;
    unsigned char(*st)[2] = stack[threadIdx.x];
    st[0][0] = (unsigned char)c; st[0][1] = (unsigned char)j; sp = 1;
    while (sp > 0) {
      if (++pops > 8 * MAXG) { bad = true; break; }             // marks never cycle in a consistent map; bounded so every wave exits
      --sp;
      const int x = st[sp][0], y = st[sp][1];
      if (x == y) continue;
      const int k = V[(long)x * G + y];
      if (k < 0) { ++hops; continue; }
      if (sp + 2 > MAXG) { bad = true; break; }
      st[sp][0] = (unsigned char)k; st[sp][1] = (unsigned char)y; ++sp;
      st[sp][0] = (unsigned char)x; st[sp][1] = (unsigned char)k; ++sp;
    }
    if (bad) { atomicExch(status, 1 + b); hops = -1; }
    f4 = xyz / 30.0;
    f5 = (c == j ? 0.0 : dis[((long)b * G + c) * G + j]) / 30.0;
    f6 = (double)hops / 10.0;
  }
  const float sh = (float)sin((double)ang_h), ch = (float)cos((double)ang_h), se = (float)sin((double)ang_e), ce = (float)cos((double)ang_e);
  for (int k = 0; k < A; k += 4) { o[k] = sh; o[k + 1] = ch; o[k + 2] = se; o[k + 3] = ce; }
  o[A] = (float)f4; o[A + 1] = (float)f5; o[A + 2] = (float)f6;
}

// grid (N, B): one output row per block
__global__ __launch_bounds__(64) void graph_pair_dists_kernel(const double* dis, const int* nodes, float* out, int G, int N) {
  const int b = blockIdx.y, i = blockIdx.x;
  const int ni = nodes[(long)b * N + i];
  float* o = out + ((long)b * N + i) * N;
  for (int j = threadIdx.x; j < N; j += 64) {
    const int nj = nodes[(long)b * N + j];
    o[j] = (ni < 0 || nj < 0 || i == j) ? 0.f : (ni == nj ? 0.f : (float)dis[((long)b * G + ni) * G + nj]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_or_zero_kernel(const T* table, long ld, const long* rows, float* out, int D) {
  const long r = rows[blockIdx.x];
  float* o = out + (long)blockIdx.x * D;
  for (int k = threadIdx.x * 4; k < D; k += 1024) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r >= 0) v = DT<T>::ld4(table + r * ld + k);
    *(f32x4*)(o + k) = v;
  }
}

}  // namespace

extern "C" int vlni_graph_init(double* dis, int* via, unsigned char* seen, int B, int G, void* stream) {
  VLNI_CHECK(dis && via && seen && B > 0 && G > 1 && G <= MAXG, VLNI_EINVAL, "graph_init: B=%d G=%d (2..%d)", B, G, MAXG);
  const long n = (long)B * G * G;
  hipLaunchKernelGGL(graph_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dis, via, seen, n, (long)B * G);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_graph_observe(double* pos, double* dis, int* via, unsigned char* seen, const int* cur, const int* cand,
                                  const double* cur_pos, const double* cand_pos, const double* cand_dist, const int* n_nodes, int B, int G, int C,
                                  void* stream) {
  VLNI_CHECK(pos && dis && via && seen && cur && n_nodes && cur_pos, VLNI_EINVAL, "graph_observe: null pointer");
  VLNI_CHECK(B > 0 && G > 1 && G <= MAXG && C >= 0 && (C == 0 || (cand && cand_pos)), VLNI_EINVAL, "graph_observe: B=%d G=%d C=%d", B, G, C);
  hipLaunchKernelGGL(graph_observe_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pos, dis, via, seen, cur, cand, cur_pos, cand_pos,
                     cand_dist, n_nodes, G, C);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_graph_pos_fts(const double* pos, const double* dis, const int* via, const int* cur, const int* nodes,
                                  const double* heading, const double* elevation, float* out, long ld_row, long ld_batch, int* status,
                                  int B, int G, int N, int A, void* stream) {
  VLNI_CHECK(pos && dis && via && cur && nodes && heading && elevation && out && status, VLNI_EINVAL, "graph_pos_fts: null pointer");
  VLNI_CHECK(B > 0 && G > 1 && G <= MAXG && N > 0 && A >= 4 && A % 4 == 0 && ld_row >= A + 3 && ld_batch >= (long)N * ld_row, VLNI_EINVAL,
             "graph_pos_fts: B=%d G=%d N=%d A=%d ld_row=%ld ld_batch=%ld", B, G, N, A, ld_row, ld_batch);
  hipLaunchKernelGGL(graph_pos_fts_kernel, dim3((N + 63) / 64, B), dim3(64), 0, (hipStream_t)stream, pos, dis, via, cur, nodes, heading,
                     elevation, out, ld_row, ld_batch, status, G, N, A);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_graph_pair_dists(const double* dis, const int* nodes, float* out, int B, int G, int N, void* stream) {
  VLNI_CHECK(dis && nodes && out && B > 0 && G > 1 && G <= MAXG && N > 0, VLNI_EINVAL, "graph_pair_dists: B=%d G=%d N=%d", B, G, N);
  hipLaunchKernelGGL(graph_pair_dists_kernel, dim3(N, B), dim3(64), 0, (hipStream_t)stream, dis, nodes, out, G, N);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}

extern "C" int vlni_gather_rows_or_zero(int dtype, const void* table, long ld, const long* rows, float* out, int n, int D, void* stream) {
  VLNI_CHECK(table && rows && out && n > 0 && D > 0 && D % 4 == 0 && ld >= D && ld % 4 == 0, VLNI_EINVAL,
             "gather_rows_or_zero: n=%d D=%d ld=%ld (D, ld multiples of 4)", n, D, ld);
  if (dtype == VLNI_F32)
    hipLaunchKernelGGL((gather_rows_or_zero_kernel<float>), dim3(n), dim3(256), 0, (hipStream_t)stream, (const float*)table, ld, rows, out, D);
  else if (dtype == VLNI_BF16)
    hipLaunchKernelGGL((gather_rows_or_zero_kernel<__bf16>), dim3(n), dim3(256), 0, (hipStream_t)stream, (const __bf16*)table, ld, rows, out, D);
  else if (dtype == VLNI_F16)
    hipLaunchKernelGGL((gather_rows_or_zero_kernel<_Float16>), dim3(n), dim3(256), 0, (hipStream_t)stream, (const _Float16*)table, ld, rows, out, D);
  else
    VLNI_CHECK(false, VLNI_EINVAL, "gather_rows_or_zero: dtype %d", dtype);
  VLNI_LAUNCH_CHECK();
  return VLNI_OK;
}
