// micro-benchmark: what would an EXPLICIT r x r Hessian per user cost?  (pcrpp.cpp:576-625)
//   H_i = lambda I + 2 X^T W,   W_p = cnt_p x_p - sum over the other levels of (range sums of the partners' rows)
// with the range sums taken from a prefix table P of the user's rows in sorted order (segment-local prefix + segment offsets),
// and X^T W accumulated by v_mfma_f32_32x32x2_f32.  One 512-thread workgroup per user, as the long classes of k_ustep.
//   k_table: one pass over the user's rows of V -> P (n x ld, global scratch)
//   k_build: one pass over the rows + (levels - 1) table rows per rating -> W chunk in LDS -> MFMA
// usage: hess_probe [nusers] [n] [r] [nitems] [levels]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int BLOCK = 512, NSEG = 16, RC = 32, LSTR = 132, MAXLEV = 5;

struct User { const int* item; const uint16_t* lev; const uint2* win; const int* rs; float* ptab; double* segoff; int n, nlev; };

__global__ __launch_bounds__(BLOCK) void k_table(const float* __restrict__ V, int ld, const int* __restrict__ item_a, float* __restrict__ ptab_a,
                                                 double* __restrict__ segoff_a, int n) {
    const int u = blockIdx.x, tid = threadIdx.x, seg = tid >> 5, c = tid & 31, nch = ld / 4;
    const int* item = item_a + (size_t)u * n;
    float* ptab = ptab_a + (size_t)u * (n + 1) * ld;
    const int seglen = (n + NSEG - 1) / NSEG, p0 = seg * seglen, p1 = min(n, p0 + seglen);
    __shared__ double tot[NSEG][128];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (c < nch) {
        for (int p = p0; p < p1; p += 8) {
            float4 x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) if (p + e < p1) x[e] = *reinterpret_cast<const float4*>(V + (size_t)item[p + e] * ld + c * 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (p + e < p1) {
                *reinterpret_cast<float4*>(ptab + (size_t)(p + e) * ld + c * 4) = float4{(float)a0, (float)a1, (float)a2, (float)a3};
                a0 += x[e].x; a1 += x[e].y; a2 += x[e].z; a3 += x[e].w;
            }
        }
        if (p1 == n && p0 <= n && n / seglen == seg)                      // row n, when it falls inside a segment
            *reinterpret_cast<float4*>(ptab + (size_t)n * ld + c * 4) = float4{(float)a0, (float)a1, (float)a2, (float)a3};
        tot[seg][c * 4] = a0; tot[seg][c * 4 + 1] = a1; tot[seg][c * 4 + 2] = a2; tot[seg][c * 4 + 3] = a3;
    }
    __syncthreads();
    // exclusive scan over the segments -> segoff[seg][t]; P(n) = the grand total is segoff[NSEG]
    for (int t = tid; t < ld; t += BLOCK) {
        double run = 0.0;
        for (int s = 0; s < NSEG; ++s) { segoff_a[((size_t)u * (NSEG + 1) + s) * ld + t] = run; run += tot[s][t]; }
        segoff_a[((size_t)u * (NSEG + 1) + NSEG) * ld + t] = run;
    }
    if (n / seglen == NSEG && tid < ld) ptab[(size_t)n * ld + tid] = 0.f;   // row n: segment NSEG, local prefix 0
}

__global__ __launch_bounds__(BLOCK) void k_build(const float* __restrict__ V, int ld, int r, const int* __restrict__ item_a,
                                                 const uint16_t* __restrict__ lev_a, const uint2* __restrict__ win_a, const int* __restrict__ rs_a,
                                                 const float* __restrict__ ptab_a, const double* __restrict__ segoff_a, int n, int nlev,
                                                 float* __restrict__ G_a, int mfma_on) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Xl = reinterpret_cast<float*>(smem);                 // [2][RC][LSTR]
    float* Wl = Xl + 2 * RC * LSTR;                             // [2][RC][LSTR]
    float* so = Wl + 2 * RC * LSTR;                             // [NSEG + 1][LSTR] segment offsets
    float* Pl = so + (NSEG + 1) * LSTR;                         // [MAXLEV + 1][LSTR]  P(rs[l])
    __shared__ int rs[MAXLEV + 1];
    const int u = blockIdx.x, tid = threadIdx.x, nch = ld / 4;
    const int* item = item_a + (size_t)u * n;
    const uint16_t* lev = lev_a + (size_t)u * n;
    const uint2* win = win_a + (size_t)u * n;
    const float* ptab = ptab_a + (size_t)u * (n + 1) * ld;
    const int seglen = (n + NSEG - 1) / NSEG;
    if (tid <= nlev) rs[tid] = rs_a[(size_t)u * (MAXLEV + 1) + tid];
    for (int i = tid; i < (NSEG + 1) * ld; i += BLOCK) so[(i / ld) * LSTR + i % ld] = (float)segoff_a[(size_t)u * (NSEG + 1) * ld + i];
    for (int i = tid; i < 2 * 2 * RC * LSTR; i += BLOCK) Xl[i] = 0.f;   // pads stay zero
    __syncthreads();
    for (int i = tid; i < (nlev + 1) * ld; i += BLOCK) {
        const int l = i / ld, t = i % ld, w = rs[l];
        Pl[l * LSTR + t] = ptab[(size_t)w * ld + t] + so[min(w / seglen, NSEG) * LSTR + t];
    }
    __syncthreads();
    // W phase mapping: 16 lanes per row, 2 rounds of 16-byte chunks
    const int prow = tid >> 4, l16 = tid & 15;
    float4 xr[2], tr[2][4]; int wi[4], lv = 0, cnt = 0;
    auto load = [&](int c0) {
        const int p = c0 * RC + prow;
        if (p >= n) { lv = -1; return; }
        const uint2 wv = win[p];
        wi[0] = wv.x & 0xFFFF; wi[1] = wv.x >> 16; wi[2] = wv.y & 0xFFFF; wi[3] = wv.y >> 16;
        lv = lev[p];
        const int it = item[p];
        cnt = 0;
#pragma unroll
        for (int l = 0; l < MAXLEV; ++l) {
            if (l >= nlev || l == lv) continue;
            if (l < lv) cnt += rs[l + 1] - wi[l < 4 ? l : 3]; else cnt += wi[l - 1 >= 0 ? l - 1 : 0] - rs[l];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = l16 + 16 * h;
            if (c >= nch) continue;
            xr[h] = *reinterpret_cast<const float4*>(V + (size_t)it * ld + c * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) if (s < nlev - 1) tr[h][s] = *reinterpret_cast<const float4*>(ptab + (size_t)wi[s] * ld + c * 4);
        }
    };
    auto store = [&](int buf) {
        float* xd = Xl + (size_t)(buf * RC + prow) * LSTR;
        float* wd = Wl + (size_t)(buf * RC + prow) * LSTR;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = l16 + 16 * h;
            if (c >= nch) continue;
            float4 w = {0, 0, 0, 0}, x = {0, 0, 0, 0};
            if (lv >= 0) {
                x = xr[h];
                w = float4{cnt * x.x, cnt * x.y, cnt * x.z, cnt * x.w};
#pragma unroll
                for (int l = 0; l < MAXLEV; ++l) {
                    if (l >= nlev || l == lv) continue;
                    const int wis = l < lv ? wi[l < 4 ? l : 3] : wi[l - 1 >= 0 ? l - 1 : 0];
                    const float4 tv = l < lv ? tr[h][l < 4 ? l : 3] : tr[h][l - 1 >= 0 ? l - 1 : 0];
                    const float4 sv = *reinterpret_cast<const float4*>(so + min(wis / seglen, NSEG) * LSTR + c * 4);
                    const float4 pw = float4{tv.x + sv.x, tv.y + sv.y, tv.z + sv.z, tv.w + sv.w};
                    const float4 pb = *reinterpret_cast<const float4*>(Pl + (l < lv ? l + 1 : l) * LSTR + c * 4);
                    const float sg = l < lv ? 1.f : -1.f;        // l < lv: P(e_l) - P(w);  l > lv: P(w) - P(s_l)
                    w.x -= sg * (pb.x - pw.x); w.y -= sg * (pb.y - pw.y); w.z -= sg * (pb.z - pw.z); w.w -= sg * (pb.w - pw.w);
                }
            }
            *reinterpret_cast<float4*>(xd + c * 4) = x;
            *reinterpret_cast<float4*>(wd + c * 4) = w;
        }
    };
    // MFMA mapping: wave w owns the 32 x 32 tiles (sb, tb0) and (sb, tb0 + 1) of G
    const int wave = tid >> 6, lane = tid & 63, sb = wave >> 1, tb0 = (wave & 1) * 2;
    f16v acc0 = {0}, acc1 = {0};
    const int nchunks = (n + RC - 1) / RC;
    load(0);
    for (int c0 = 0; c0 < nchunks; ++c0) {
        const int buf = c0 & 1;
        store(buf);
        __syncthreads();
        if (c0 + 1 < nchunks) load(c0 + 1);
        if (mfma_on) {
            const float* xa = Xl + (size_t)(buf * RC + (lane >> 5)) * LSTR + sb * 32 + (lane & 31);
            const float* wb = Wl + (size_t)(buf * RC + (lane >> 5)) * LSTR + tb0 * 32 + (lane & 31);
#pragma unroll
            for (int k = 0; k < RC; k += 2) {
                const float a = xa[k * LSTR], b0 = wb[k * LSTR], b1 = wb[k * LSTR + 32];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            }
        }
    }
    float* G = G_a + (size_t)u * 128 * 128;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = sb * 32 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5), col = lane & 31;
        G[row * 128 + tb0 * 32 + col] = acc0[j];
        G[row * 128 + (tb0 + 1) * 32 + col] = acc1[j];
    }
}

int main(int argc, char** argv) {
    const int nusers = argc > 1 ? atoi(argv[1]) : 512, n = argc > 2 ? atoi(argv[2]) : 4096, r = argc > 3 ? atoi(argv[3]) : 100;
    const int nitems = argc > 4 ? atoi(argv[4]) : 17770, nlev = argc > 5 ? atoi(argv[5]) : 5;
    const int ld = (r + 3) / 4 * 4;
    if (r > 128 || nlev > MAXLEV || n > 65535) { printf("r <= 128, levels <= 5, n < 65536\n"); return 1; }
    srand(7);
    std::vector<float> V((size_t)nitems * ld);
    for (auto& v : V) v = (rand() % 2001 - 1000) * 1e-3f;
    std::vector<int> item((size_t)nusers * n), rs((size_t)nusers * (MAXLEV + 1));
    std::vector<uint16_t> lev((size_t)nusers * n), win((size_t)nusers * n * 4);
    for (int u = 0; u < nusers; ++u) {
        for (int l = 0; l <= nlev; ++l) rs[(size_t)u * (MAXLEV + 1) + l] = (int)((long)n * l / nlev);
        const int* R = &rs[(size_t)u * (MAXLEV + 1)];
        for (int p = 0; p < n; ++p) {
            item[(size_t)u * n + p] = rand() % nitems;
            int lv = 0; while (p >= R[lv + 1]) ++lv;
            lev[(size_t)u * n + p] = (uint16_t)lv;
            const double f = (double)(p - R[lv]) / (R[lv + 1] - R[lv]);
            for (int l = 0; l < nlev; ++l) {
                if (l == lv) continue;
                // monotone boundaries: lower levels keep a suffix that shrinks as m grows, higher levels a prefix that grows
                double g = f + (l < lv ? -0.15 : 0.15); g = g < 0 ? 0 : g > 1 ? 1 : g;
                const int w = R[l] + (int)(g * (R[l + 1] - R[l]));
                win[((size_t)u * n + p) * 4 + (l < lv ? l : l - 1)] = (uint16_t)w;
            }
        }
    }
    float *dV, *dP, *dG; int *di, *drs; uint16_t *dl, *dw; double* dso;
    CK(hipMalloc(&dV, V.size() * 4)); CK(hipMalloc(&di, item.size() * 4)); CK(hipMalloc(&drs, rs.size() * 4));
    CK(hipMalloc(&dl, lev.size() * 2)); CK(hipMalloc(&dw, win.size() * 2));
    CK(hipMalloc(&dP, (size_t)nusers * (n + 1) * ld * 4)); CK(hipMalloc(&dso, (size_t)nusers * (NSEG + 1) * ld * 8));
    CK(hipMalloc(&dG, (size_t)nusers * 128 * 128 * 4));
    CK(hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(di, item.data(), item.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drs, rs.data(), rs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, lev.data(), lev.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, win.data(), win.size() * 2, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(4 * RC * LSTR + (NSEG + 1) * LSTR + (MAXLEV + 1) * LSTR) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto&& f) {
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-28s %9.1f us   %7.2f us per user per CU-slot (256)   %6.2f ns per rating\n", name, ms * 1e3, ms * 1e3 / ((nusers + 255) / 256),
               ms * 1e6 / ((double)nusers * n));
    };
    timeit("k_table", [&] { hipLaunchKernelGGL(k_table, dim3(nusers), dim3(BLOCK), 0, 0, dV, ld, di, dP, dso, n); });
    timeit("k_build (no MFMA)", [&] { hipLaunchKernelGGL(k_build, dim3(nusers), dim3(BLOCK), lds, 0, dV, ld, r, di, dl, reinterpret_cast<const uint2*>(dw), drs, dP, dso, n, nlev, dG, 0); });
    timeit("k_build", [&] { hipLaunchKernelGGL(k_build, dim3(nusers), dim3(BLOCK), lds, 0, dV, ld, r, di, dl, reinterpret_cast<const uint2*>(dw), drs, dP, dso, n, nlev, dG, 1); });
    CK(hipGetLastError());
    // check user 0 against a double-precision host computation
    std::vector<float> G(128 * 128);
    CK(hipMemcpy(G.data(), dG, G.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> P((size_t)(n + 1) * ld, 0.0), Gh((size_t)ld * ld, 0.0), W(ld);
    for (int p = 0; p < n; ++p) for (int t = 0; t < ld; ++t) P[(size_t)(p + 1) * ld + t] = P[(size_t)p * ld + t] + V[(size_t)item[p] * ld + t];
    for (int p = 0; p < n; ++p) {
        const int lv = lev[p]; int cnt = 0;
        for (int t = 0; t < ld; ++t) W[t] = 0.0;
        for (int l = 0; l < nlev; ++l) {
            if (l == lv) continue;
            const int w = win[(size_t)p * 4 + (l < lv ? l : l - 1)];
            const int a = l < lv ? w : rs[l], b = l < lv ? rs[l + 1] : w;
            cnt += b - a;
            for (int t = 0; t < ld; ++t) W[t] -= P[(size_t)b * ld + t] - P[(size_t)a * ld + t];
        }
        const float* x = &V[(size_t)item[p] * ld];
        for (int t = 0; t < ld; ++t) W[t] += (double)cnt * x[t];
        for (int s = 0; s < ld; ++s) for (int t = 0; t < ld; ++t) Gh[(size_t)s * ld + t] += (double)x[s] * W[t];
    }
    double maxrel = 0.0, nrm = 0.0;
    for (int s = 0; s < ld; ++s) for (int t = 0; t < ld; ++t) nrm = fmax(nrm, fabs(Gh[(size_t)s * ld + t]));
    for (int s = 0; s < ld; ++s) for (int t = 0; t < ld; ++t) maxrel = fmax(maxrel, fabs(G[s * 128 + t] - Gh[(size_t)s * ld + t]) / nrm);
    printf("user 0: max |G - G_host| / max|G| = %.3g  (fp32 table, fp32 MFMA accumulation over %d rows)\n", maxrel, n);
    return 0;
}
