// On-device evaluation of the moment tasks: the interval arithmetic of the reference's evaluate.py on
// batches of videos, in double precision with Python's operation order, so the decisions (IoU vs threshold, NMS
// suppression) are the reference's decisions bit for bit.  Tiny integer/branchy work: one thread per interval pair or
// per video; no MFMA, no LDS.
#include "common.h"

namespace {

// evaluate.py:24-31 compute_iou(interval_1 = (s_i, e_i), interval_2 = (s, e))
__device__ __forceinline__ double iou_py(double s_i, double e_i, double s, double e) {
    const double inter = fmax(0.0, fmin(e, e_i) - fmax(s, s_i));
    const double uni = fmin(fmax(e, e_i) - fmin(s, s_i), ((e - s) + e_i) - s_i);   // Python: end-start + end_i-start_i
    return inter / (uni + 1e-8);
}

__global__ void interval_iou_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n, double* __restrict__ iou) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) iou[i] = iou_py(a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1]);
}

// evaluate.py:123-188, body of the per-video loop: pred p covers ref r when compute_iou(pred, ref) > tiou (strict)
__global__ void step_bound_pr_kernel(const double* __restrict__ refs, const int32_t* __restrict__ ref_off,
                                     const double* __restrict__ preds, const int32_t* __restrict__ pred_off, int V, double tiou,
                                     double* __restrict__ recall, double* __restrict__ precision, double* __restrict__ best_iou) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int r0 = ref_off[v], r1 = ref_off[v + 1], p0 = pred_off[v], p1 = pred_off[v + 1];
    int pred_cov = 0, ref_cov = 0;
    for (int p = p0; p < p1; ++p) {
        bool any = false;
        double best = -1.0;
        for (int r = r0; r < r1; ++r) {
            const double iu = iou_py(preds[2 * p], preds[2 * p + 1], refs[2 * r], refs[2 * r + 1]);
            any |= iu > tiou;
            best = iu > best ? iu : best;
        }
        pred_cov += any;
        if (best_iou) best_iou[p] = best;                        // the `ious` lists of :158-159 (max over refs)
    }
    for (int r = r0; r < r1; ++r) {
        bool any = false;
        for (int p = p0; p < p1; ++p) any |= iou_py(preds[2 * p], preds[2 * p + 1], refs[2 * r], refs[2 * r + 1]) > tiou;
        ref_cov += any;
    }
    precision[v] = (double)pred_cov / (double)(p1 - p0);         // len(pred_set_covered) / (pred_i + 1)
    recall[v] = (double)ref_cov / (double)(r1 - r0);
}

// evaluate.py:322-412 preprocess_moment_bounds for one video per thread:
//   keep predictions strictly inside (min_x, max_x); NMS(overlapThresh = 0) over boxes [x1, 0, x2, 1] visiting candidates
//   from the last index down (np.argsort of the all-equal y2 is the identity); sort survivors by start (stable);
//   fill every gap, including before the first and after the last survivor.
constexpr int MAXP = 128;
__global__ void preprocess_bounds_kernel(const double* __restrict__ preds, const int32_t* __restrict__ pred_off,
                                         const double* __restrict__ gt_minmax, int V, double* __restrict__ out,
                                         int32_t* __restrict__ out_count, int max_out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const double min_x = gt_minmax[2 * v], max_x = gt_minmax[2 * v + 1];
    double x1[MAXP], x2[MAXP];
    int idx[MAXP];
    int n = 0;
    for (int p = pred_off[v]; p < pred_off[v + 1] && n < MAXP; ++p) {
        const double a = preds[2 * p], b = preds[2 * p + 1];
        if (a > min_x && b < max_x) { x1[n] = a; x2[n] = b; idx[n] = n; ++n; }
    }
    double s1[MAXP], s2[MAXP];
    int npick = 0, m = n;
    while (m > 0) {
        const int i = idx[m - 1];
        s1[npick] = x1[i]; s2[npick] = x2[i]; ++npick;
        int w = 0;
        for (int t = 0; t < m - 1; ++t) {
            const int j = idx[t];
            const double xx1 = fmax(x1[i], x1[j]), xx2 = fmin(x2[i], x2[j]);
            const double ww = fmax(0.0, xx2 - xx1 + 1.0), hh = fmax(0.0, 1.0 - 0.0 + 1.0);
            const double area = (x2[j] - x1[j] + 1.0) * (1.0 - 0.0 + 1.0);
            const double overlap = (ww * hh) / area;
            if (!(overlap > 0.0)) idx[w++] = j;                  // np.where(overlap > overlapThresh) are deleted
        }
        m = w;
    }
    double* o = out + (int64_t)v * max_out * 2;
    int cnt = 0;
    auto emit = [&](double a, double b) { if (cnt < max_out) { o[2 * cnt] = a; o[2 * cnt + 1] = b; } ++cnt; };
    if (npick == 0) {
        emit(min_x, max_x);
    } else {
        for (int a = 1; a < npick; ++a) {                        // stable insertion sort by start (list.sort(key=x[0]))
            const double k1 = s1[a], k2 = s2[a];
            int b = a - 1;
            while (b >= 0 && s1[b] > k1) { s1[b + 1] = s1[b]; s2[b + 1] = s2[b]; --b; }
            s1[b + 1] = k1; s2[b + 1] = k2;
        }
        if (s1[0] > min_x) emit(min_x, s1[0]);
        double last_end = 0.0;
        for (int a = 0; a < npick; ++a) {
            emit(s1[a], s2[a]);
            last_end = s2[a];
            if (a + 1 < npick) { emit(s2[a], s1[a + 1]); last_end = s1[a + 1]; }
        }
        if (last_end < max_x) emit(last_end, max_x);
    }
    out_count[v] = cnt;
}

// hirest_dataset.py:12-68: bins = np.linspace(0, int(duration) - 1, n).  numpy builds it as arange(n) * step with
// step = (stop - start) / (n - 1) in double and overwrites the last element with `stop`; the bin value is recomputed
// here on demand (one multiply) instead of materialising n doubles per conversion as the reference does.
struct Bins {
    int64_t n; double stop, step;
    __device__ __forceinline__ double at(int64_t i) const { return (i == n - 1 && n > 1) ? stop : (n > 1 ? (double)i * step : 0.0); }
};
__device__ __forceinline__ bool make_bins(double duration, int32_t n_frames, Bins& b) {
    const int64_t d = (int64_t)duration;                       // Python int(): truncation toward zero
    b.n = n_frames < 0 ? d : n_frames;                         // n_frames < 0: one frame per second
    b.stop = (double)(d - 1);
    b.step = b.n > 1 ? b.stop / (double)(b.n - 1) : 0.0;
    return d >= 1 && b.n >= 1;                                 // shorter than one second: the reference's bins are empty / decreasing
}

__global__ void frame_to_timestamp_kernel(const int64_t* __restrict__ frame, const double* __restrict__ duration,
                                          const int32_t* __restrict__ n_frames, int32_t n_frames_all, int64_t per_video, int64_t n,
                                          int64_t* __restrict__ ts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = i / per_video;
    Bins b;
    int64_t f = frame[i];
    if (!make_bins(duration[v], n_frames ? n_frames[v] : n_frames_all, b)) { ts[i] = INT64_MIN; return; }
    if (f < 0) f += b.n;                                       // numpy negative indexing
    ts[i] = (f < 0 || f >= b.n) ? INT64_MIN : (int64_t)b.at(f);   // IndexError in the reference
}

// np.digitize(t, bins, right=True) = number of bins strictly below t, then min(., n - 1)
__global__ void timestamp_to_frame_kernel(const double* __restrict__ t, const double* __restrict__ duration,
                                          const int32_t* __restrict__ n_frames, int32_t n_frames_all, int64_t per_video, int64_t n,
                                          int64_t* __restrict__ frame) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = i / per_video;
    Bins b;
    if (!make_bins(duration[v], n_frames ? n_frames[v] : n_frames_all, b)) { frame[i] = INT64_MIN; return; }
    const double x = t[i];
    if (x != x) { frame[i] = b.n - 1; return; }                // NaN sorts after every bin
    int64_t k = 0;
    if (b.step > 0.0 && x > 0.0) {                             // first guess from the spacing, then settle on the exact bin values
        const double g = ceil(x / b.step);
        k = g >= (double)b.n ? b.n : (int64_t)g;
    }
    while (k > 0 && !(b.at(k - 1) < x)) --k;
    while (k < b.n && b.at(k) < x) ++k;
    frame[i] = k < b.n - 1 ? k : b.n - 1;
}

}  // namespace

extern "C" int hirest_interval_iou_f64(const double* a, const double* b, int64_t n, double* iou, void* stream) {
    if (n == 0) return 0;
    if (!a || !b || !iou || n < 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(interval_iou_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, iou);
    return hirest_launch_status();
}

extern "C" int hirest_step_bound_pr(const double* refs, const int32_t* ref_off, const double* preds, const int32_t* pred_off,
                                    int32_t V, double tiou, double* recall, double* precision, double* best_iou, void* stream) {
    if (V == 0) return 0;
    if (!refs || !ref_off || !preds || !pred_off || !recall || !precision || V < 0) return HIREST_E_BADARG;
    hipLaunchKernelGGL(step_bound_pr_kernel, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, refs, ref_off, preds, pred_off,
                       V, tiou, recall, precision, best_iou);
    return hirest_launch_status();
}

extern "C" int hirest_preprocess_moment_bounds(const double* preds, const int32_t* pred_off, const double* gt_minmax, int32_t V,
                                               double* out, int32_t* out_count, int32_t max_out, void* stream) {
    if (V == 0) return 0;
    if (!preds || !pred_off || !gt_minmax || !out || !out_count || V < 0 || max_out < 1) return HIREST_E_BADARG;
    hipLaunchKernelGGL(preprocess_bounds_kernel, dim3((V + 63) / 64), dim3(64), 0, (hipStream_t)stream, preds, pred_off, gt_minmax,
                       V, out, out_count, max_out);
    return hirest_launch_status();
}

extern "C" int hirest_frame_to_timestamp(const int64_t* frame, const double* duration, const int32_t* n_frames, int32_t n_frames_all,
                                         int64_t per_video, int64_t n, int64_t* timestamp, void* stream) {
    if (n == 0) return 0;
    if (!frame || !duration || !timestamp || n < 0 || per_video < 1) return HIREST_E_BADARG;
    hipLaunchKernelGGL(frame_to_timestamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frame, duration,
                       n_frames, n_frames_all, per_video, n, timestamp);
    return hirest_launch_status();
}

extern "C" int hirest_timestamp_to_frame(const double* t, const double* duration, const int32_t* n_frames, int32_t n_frames_all,
                                         int64_t per_video, int64_t n, int64_t* frame, void* stream) {
    if (n == 0) return 0;
    if (!t || !duration || !frame || n < 0 || per_video < 1) return HIREST_E_BADARG;
    hipLaunchKernelGGL(timestamp_to_frame_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, duration,
                       n_frames, n_frames_all, per_video, n, frame);
    return hirest_launch_status();
}
