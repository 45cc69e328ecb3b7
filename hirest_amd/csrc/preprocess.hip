// Frame preprocessing for gfx950: Pillow-exact antialiased bicubic resize (short side -> size) + centre crop on
// uint8 RGB frames, optional ToTensor+Normalize.  Replaces the single-threaded PIL pipeline that bounds the
// reference's feature extraction (eva_clip.py:125-153; extract_features.py:46-50).  Integer byte work, HBM-bound
// on the input read: no MFMA here.
//
// Layout.  frames [B,H,W,3] u8.  Pass 1 (horizontal) resamples only the input rows the cropped vertical pass
// needs and only the cropped columns: tmp [B, nrows, size, 3] u8.  Pass 2 (vertical) produces the crop.
// Plan blob (int32 words): hdr[16] | hb[size][2] | hk[ksh][size] | vb[size][2] | vk[size][ksv]
// (hk is tap-major so that lanes = output columns read consecutive words; vk rows are wave-uniform).
#include "common.h"
#include <math.h>

#pragma clang fp contract(off)   // the weight tables must round exactly like Pillow's C doubles

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow Resample.c

struct Geometry {
    int nw, nh, left, top, ksh, ksv, row0, nrows, col0, ncols;
};

inline double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

inline int ksize_for(int in_size, int out_size) {
    if (in_size == out_size) return 1;                    // Pillow skips a pass that keeps the size
    double scale = (double)in_size / out_size;
    double fs = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * fs) * 2 + 1;
}

// Bounds / fixed-point weights of output index xx (Resample.c precompute_coeffs + normalize_coeffs_8bpc).
// k must hold ksize ints; returns (xmin, count).
inline void coeffs_for(int in_size, int out_size, int xx, int ksize, int* xmin_out, int* n_out, int32_t* k) {
    for (int i = 0; i < ksize; ++i) k[i] = 0;
    if (in_size == out_size) { *xmin_out = xx; *n_out = 1; k[0] = 1 << PRECISION_BITS; return; }
    double scale = (double)in_size / out_size;
    double fs = scale < 1.0 ? 1.0 : scale;
    double support = 2.0 * fs;
    double ss = 1.0 / fs;
    double center = 0.0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double w[512];
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) { w[x] = bicubic((x + xmin - center + 0.5) * ss); ww += w[x]; }
    for (int x = 0; x < xmax; ++x) {
        double v = ww != 0.0 ? w[x] / ww : w[x];
        k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    *xmin_out = xmin; *n_out = xmax;
}

// torchvision Resize(int) + CenterCrop(int) bookkeeping.
inline long round_half_even(double v) { return lrint(v); }   // default rounding mode = to nearest even, like Python round()
inline int geometry(int in_h, int in_w, int size, Geometry* g) {
    if (in_h < 1 || in_w < 1 || size < 1) return HIREST_E_BADARG;
    int w = in_w, h = in_h;
    if ((w <= h && w == size) || (h <= w && h == size)) { g->nw = w; g->nh = h; }
    else if (w < h) { g->nw = size; g->nh = (int)((double)size * h / w); }
    else { g->nh = size; g->nw = (int)((double)size * w / h); }
    g->left = (int)round_half_even((g->nw - size) / 2.0);
    g->top = (int)round_half_even((g->nh - size) / 2.0);
    if (g->left < 0 || g->top < 0) return HIREST_E_BADARG;
    g->ksh = ksize_for(in_w, g->nw);
    g->ksv = ksize_for(in_h, g->nh);
    if (g->ksh > 512 || g->ksv > 512) return HIREST_E_BADARG;
    int32_t k[512]; int lo, n, lo2, n2;
    coeffs_for(in_h, g->nh, g->top, g->ksv, &lo, &n, k);
    coeffs_for(in_h, g->nh, g->top + size - 1, g->ksv, &lo2, &n2, k);
    g->row0 = lo; g->nrows = lo2 + n2 - lo;
    coeffs_for(in_w, g->nw, g->left, g->ksh, &lo, &n, k);
    coeffs_for(in_w, g->nw, g->left + size - 1, g->ksh, &lo2, &n2, k);
    g->col0 = lo; g->ncols = lo2 + n2 - lo;
    return 0;
}

inline int64_t plan_words(const Geometry& g, int size) {
    return 16 + (int64_t)size * 2 + (int64_t)g.ksh * size + (int64_t)size * 2 + (int64_t)size * g.ksv;
}

// ---- pass 1: horizontal.  One block per (band of H_ROWS input rows, frame); lanes = output columns.
// The weight table is staged in LDS once per block; input rows are double-buffered (row i+1 is in flight in
// registers while row i is resampled), so a block's global latency is paid once, not once per tap.
constexpr int H_ROWS = 8;
constexpr int H_MAXW = 12;                       // row dwords per thread: rows up to 256*12*4 = 12288 B (4096 px)

__device__ __forceinline__ uint32_t row_word(const uint8_t* rowp, const uint32_t* ap, int i, int nd, int mis, int nbytes) {
    if (i == 0 || i == nd - 1) {                 // edge words: never touch bytes outside this row's span
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            int idx = i * 4 + k - mis;
            if (idx >= 0 && idx < nbytes) v |= (uint32_t)rowp[idx] << (8 * k);
        }
        return v;
    }
    return ap[i];
}

__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ frames, int in_h, int in_w, int size,
                                                         int row0, int nrows, int col0, int ncols, int ksh,
                                                         const int32_t* __restrict__ hb, const int32_t* __restrict__ hk,
                                                         uint8_t* __restrict__ tmp, int row_words) {
    extern __shared__ uint32_t lds32[];
    int32_t* coef = (int32_t*)lds32;                                   // [ksh][size]
    uint32_t* rowbuf = lds32 + ksh * size;                             // [2][row_words]
    const int b = blockIdx.y, tid = threadIdx.x;
    const int r_begin = blockIdx.x * H_ROWS, r_end = min(r_begin + H_ROWS, nrows);
    const int nbytes = ncols * 3;
    for (int i = tid; i < ksh * size; i += 256) coef[i] = hk[i];

    uint32_t v[H_MAXW] = {};
    auto fetch = [&](int r) {
        const uint8_t* rowp = frames + (((size_t)b * in_h + row0 + r) * in_w + col0) * 3;
        const int mis = (int)((uintptr_t)rowp & 3);
        const uint32_t* ap = (const uint32_t*)(rowp - mis);
        const int nd = (mis + nbytes + 3) >> 2;
#pragma unroll
        for (int j = 0; j < H_MAXW; ++j) {
            const int i = tid + j * 256;
            if (i < nd) v[j] = row_word(rowp, ap, i, nd, mis, nbytes);
        }
        return mis;
    };
    auto stash = [&](uint32_t* dst) {
#pragma unroll
        for (int j = 0; j < H_MAXW; ++j) {
            const int i = tid + j * 256;
            if (i < row_words) dst[i] = v[j];
        }
    };
    int mis = fetch(r_begin);
    stash(rowbuf);
    __syncthreads();
    for (int r = r_begin; r < r_end; ++r) {
        const int cur = (r - r_begin) & 1;
        int mis_next = 0;
        if (r + 1 < r_end) mis_next = fetch(r + 1);
        const uint8_t* row = (const uint8_t*)(rowbuf + cur * row_words) + mis;
        for (int x = tid; x < size; x += 256) {
            const int lo = hb[2 * x] - col0, n = hb[2 * x + 1];
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
            const uint8_t* p = row + lo * 3;
#pragma unroll 4
            for (int t = 0; t < n; ++t) {
                const int k = coef[t * size + x];
                a0 += (int)p[3 * t] * k; a1 += (int)p[3 * t + 1] * k; a2 += (int)p[3 * t + 2] * k;
            }
            uint8_t* o = tmp + (((size_t)b * nrows + r) * size + x) * 3;
            o[0] = (uint8_t)min(max(a0 >> PRECISION_BITS, 0), 255);
            o[1] = (uint8_t)min(max(a1 >> PRECISION_BITS, 0), 255);
            o[2] = (uint8_t)min(max(a2 >> PRECISION_BITS, 0), 255);
        }
        if (r + 1 < r_end) stash(rowbuf + (cur ^ 1) * row_words);
        mis = mis_next;
        __syncthreads();
    }
}

// ---- pass 2: vertical.  One block per (band of V_ROWS output rows, frame). ----
constexpr int V_ROWS = 8;

template <int KIND>
__device__ __forceinline__ void emit(void* out, int b, int y, int j, int size, int v, const float* mean3, const float* std3) {
    if (KIND == 0) {
        ((uint8_t*)out)[((size_t)b * size + y) * size * 3 + j] = (uint8_t)v;
    } else {
        const int x = j / 3, c = j - 3 * x;
        ((float*)out)[(((size_t)b * 3 + c) * size + y) * size + x] = ((float)v / 255.0f - mean3[c]) / std3[c];
    }
}

// word path: lanes = 4 consecutive bytes of a resampled row (row bytes % 4 == 0, workspace 4-byte aligned)
template <int KIND>
__global__ __launch_bounds__(256) void resample_v_kernel_w(const uint32_t* __restrict__ tmp, int size, int row0, int nrows, int ksv,
                                                           const int32_t* __restrict__ vb, const int32_t* __restrict__ vk,
                                                           void* __restrict__ out, const float* __restrict__ mean3,
                                                           const float* __restrict__ std3) {
    const int b = blockIdx.y;
    const int wpr = size * 3 / 4;
    const int y_begin = blockIdx.x * V_ROWS, ny = min(V_ROWS, size - y_begin);
    for (int item = threadIdx.x; item < ny * wpr; item += 256) {
        const int y = y_begin + item / wpr, j = item % wpr;
        const int lo = vb[2 * y] - row0, n = vb[2 * y + 1];
        const uint32_t* src = tmp + ((size_t)b * nrows + lo) * wpr + j;
        const int32_t* k = vk + (size_t)y * ksv;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
#pragma unroll 4
        for (int t = 0; t < n; ++t) {
            const uint32_t w = src[(size_t)t * wpr];
            const int kt = k[t];
            a0 += (int)(w & 255) * kt; a1 += (int)((w >> 8) & 255) * kt;
            a2 += (int)((w >> 16) & 255) * kt; a3 += (int)(w >> 24) * kt;
        }
        const int v0 = min(max(a0 >> PRECISION_BITS, 0), 255), v1 = min(max(a1 >> PRECISION_BITS, 0), 255);
        const int v2 = min(max(a2 >> PRECISION_BITS, 0), 255), v3 = min(max(a3 >> PRECISION_BITS, 0), 255);
        if (KIND == 0) {
            ((uint32_t*)out)[((size_t)b * size + y) * wpr + j] = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
        } else {
            emit<1>(out, b, y, 4 * j, size, v0, mean3, std3); emit<1>(out, b, y, 4 * j + 1, size, v1, mean3, std3);
            emit<1>(out, b, y, 4 * j + 2, size, v2, mean3, std3); emit<1>(out, b, y, 4 * j + 3, size, v3, mean3, std3);
        }
    }
}

// byte path for sizes whose rows are not whole words
template <int KIND>
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, int size, int row0, int nrows, int ksv,
                                                         const int32_t* __restrict__ vb, const int32_t* __restrict__ vk,
                                                         void* __restrict__ out, const float* __restrict__ mean3,
                                                         const float* __restrict__ std3) {
    const int y = blockIdx.x, b = blockIdx.y;
    const int lo = vb[2 * y] - row0, n = vb[2 * y + 1];
    const int rb = size * 3;
    const uint8_t* src = tmp + ((size_t)b * nrows + lo) * rb;
    const int32_t* k = vk + (size_t)y * ksv;
    for (int j = threadIdx.x; j < rb; j += 256) {
        int acc = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) acc += (int)src[(size_t)t * rb + j] * k[t];
        emit<KIND>(out, b, y, j, size, min(max(acc >> PRECISION_BITS, 0), 255), mean3, std3);
    }
}

}  // namespace

extern "C" int64_t hirest_preprocess_plan_bytes(int32_t in_h, int32_t in_w, int32_t size) {
    Geometry g;
    if (geometry(in_h, in_w, size, &g)) return HIREST_E_BADARG;
    return plan_words(g, size) * 4;
}

extern "C" int hirest_preprocess_plan(int32_t in_h, int32_t in_w, int32_t size, void* host_plan, int64_t plan_bytes) {
    Geometry g;
    if (!host_plan) return HIREST_E_BADARG;
    if (int rc = geometry(in_h, in_w, size, &g)) return rc;
    if (plan_bytes < plan_words(g, size) * 4) return HIREST_E_WORKSPACE;
    int32_t* w = (int32_t*)host_plan;
    int32_t hdr[16] = {in_h, in_w, size, g.nw, g.nh, g.left, g.top, g.ksh, g.ksv, g.row0, g.nrows, g.col0, g.ncols, 0, 0, 0};
    for (int i = 0; i < 16; ++i) w[i] = hdr[i];
    int32_t* hb = w + 16;
    int32_t* hk = hb + 2 * size;
    int32_t* vb = hk + (int64_t)g.ksh * size;
    int32_t* vk = vb + 2 * size;
    int32_t k[512];
    for (int x = 0; x < size; ++x) {
        int lo, n;
        coeffs_for(in_w, g.nw, g.left + x, g.ksh, &lo, &n, k);
        hb[2 * x] = lo; hb[2 * x + 1] = n;
        for (int t = 0; t < g.ksh; ++t) hk[(int64_t)t * size + x] = k[t];
        coeffs_for(in_h, g.nh, g.top + x, g.ksv, &lo, &n, k);
        vb[2 * x] = lo; vb[2 * x + 1] = n;
        for (int t = 0; t < g.ksv; ++t) vk[(int64_t)x * g.ksv + t] = k[t];
    }
    return 0;
}

extern "C" int64_t hirest_preprocess_workspace_bytes(int32_t in_h, int32_t in_w, int32_t size, int32_t B) {
    Geometry g;
    if (B < 0 || geometry(in_h, in_w, size, &g)) return HIREST_E_BADARG;
    return (int64_t)B * g.nrows * size * 3;
}

extern "C" int hirest_preprocess_u8(const uint8_t* frames, int32_t B, int32_t in_h, int32_t in_w, int32_t size,
                                    const void* plan_dev, void* out, int32_t out_kind, const float* mean3, const float* std3,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
    Geometry g;
    if (B == 0) return 0;
    if (!frames || !plan_dev || !out || !workspace || B < 0 || B > 65535) return HIREST_E_BADARG;
    if (out_kind != 0 && out_kind != 1) return HIREST_E_BADARG;
    if (out_kind == 1 && (!mean3 || !std3)) return HIREST_E_BADARG;
    if (int rc = geometry(in_h, in_w, size, &g)) return rc;
    if (workspace_bytes < (int64_t)B * g.nrows * size * 3) return HIREST_E_WORKSPACE;
    const int32_t* w = (const int32_t*)plan_dev;
    const int32_t* hb = w + 16;
    const int32_t* hk = hb + 2 * size;
    const int32_t* vb = hk + (int64_t)g.ksh * size;
    const int32_t* vk = vb + 2 * size;
    hipStream_t s = (hipStream_t)stream;
    const int row_words = (g.ncols * 3 + 3 + 3) / 4 + 1;
    if (row_words > 256 * H_MAXW) return HIREST_E_SHAPE;                     // frames wider than 4096 px
    const size_t lds = ((size_t)g.ksh * size + 2 * (size_t)row_words) * 4;
    if (lds > 64 * 1024) return HIREST_E_SHAPE;
    hipLaunchKernelGGL(resample_h_kernel, dim3((g.nrows + H_ROWS - 1) / H_ROWS, B), dim3(256), lds, s, frames, in_h, in_w, size,
                       g.row0, g.nrows, g.col0, g.ncols, g.ksh, hb, hk, (uint8_t*)workspace, row_words);
    const bool words = (size * 3) % 4 == 0 && ((uintptr_t)workspace & 3) == 0 && ((uintptr_t)out & 3) == 0;
    if (words) {
        const dim3 grid((size + V_ROWS - 1) / V_ROWS, B);
        if (out_kind == 0)
            hipLaunchKernelGGL(resample_v_kernel_w<0>, grid, dim3(256), 0, s, (const uint32_t*)workspace, size, g.row0, g.nrows,
                               g.ksv, vb, vk, out, mean3, std3);
        else
            hipLaunchKernelGGL(resample_v_kernel_w<1>, grid, dim3(256), 0, s, (const uint32_t*)workspace, size, g.row0, g.nrows,
                               g.ksv, vb, vk, out, mean3, std3);
    } else if (out_kind == 0) {
        hipLaunchKernelGGL(resample_v_kernel<0>, dim3(size, B), dim3(256), 0, s, (const uint8_t*)workspace, size, g.row0,
                           g.nrows, g.ksv, vb, vk, out, mean3, std3);
    } else {
        hipLaunchKernelGGL(resample_v_kernel<1>, dim3(size, B), dim3(256), 0, s, (const uint8_t*)workspace, size, g.row0,
                           g.nrows, g.ksv, vb, vk, out, mean3, std3);
    }
    return hirest_launch_status();
}
