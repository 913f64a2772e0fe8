// On-device input pipeline in front of the backbone: decoded uint8 frames / label maps in HBM ->
// cropped, resized, flipped, normalised network inputs, bit-identical to what the reference's loader produces
// with PIL + torchvision on the host (BEV/Dataloader/Load_Data_new.py:62-117, BP/Dataloader/Load_Data_new.py:
// 110-197):   F.crop(bottom 640 rows) -> F.resize((R, 2R), BILINEAR | NEAREST) -> F.hflip -> ToTensor.
//
// The resampling arithmetic is Pillow's (libImaging/Resample.c, Geometry.c; not part of /root/reference):
//   BILINEAR = separable triangle filter widened by the down-scale factor, coefficients normalised in double and
//   rounded to 22-bit fixed point, horizontal pass -> clip to uint8 -> vertical pass -> clip to uint8;
//   NEAREST  = source index from a coordinate ACCUMULATED in double (xo += scale), truncated.
// Both tables are built on the host exactly that way (plan), the kernels only do integer work.
//
// Byte-oriented and HBM-bound (2.4 MB in, 1.5 MB out per 720x1280 frame at R = 256): each workgroup stages the
// input window of an output tile (16 x 64 at the usual 2-2.5x down-scale) in LDS once, runs the horizontal pass LDS -> LDS and the vertical pass
// LDS -> registers, and writes fp32 NCHW rows of 64 consecutive pixels.
#include <math.h>
#include <stdint.h>

#include <vector>

#include "lf_common.h"

#define PIL_PRECISION_BITS 22
#define TILE_H_MAX 16
#define TILE_W_MAX 64
#define LDS_BUDGET (48 * 1024)

struct lf_pipeline_plan {
    int Hin, Win, crop_top, crop_h, out_h, out_w;
    int ksx, ksy;
    std::vector<int> bx, kx, by, ky;     // bilinear: bounds (first, count) and fixed-point weights
    std::vector<int> ntx, nty;           // nearest source index tables
    int tile_h, tile_w;                  // output tile per workgroup (shrunk until its input window fits the LDS budget)
    int max_rows, max_cols;              // largest input window of an output tile
    size_t lds_bytes;
    int in_stride;                       // LDS bytes per staged input row (dword aligned, + shift + tap over-read pad)
    long table_ints;
};

namespace {

// Pillow precompute_coeffs (triangle filter, support 1.0) + normalize_coeffs_8bpc over a whole axis
int pil_bilinear_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk) {
    double scale, filterscale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w[x] = a < 1.0 ? 1.0 - a : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            kk[(size_t)xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PIL_PRECISION_BITS))
                                                  : (int)(0.5 + w[x] * (1 << PIL_PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

// Pillow ImagingScaleAffine, nearest filter
void pil_nearest_table(int in_size, int out_size, std::vector<int>& tab) {
    const double a = (double)in_size / out_size;
    double xo = a * 0.5;
    tab.resize(out_size);
    for (int x = 0; x < out_size; ++x) {
        int xin = xo < 0.0 ? -1 : (int)xo;
        if (xin < 0) xin = 0;
        if (xin > in_size - 1) xin = in_size - 1;
        tab[x] = xin;
        xo += a;
    }
}

__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> PIL_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// device table block layout (ints): bx[2*ow] kx[ow*ksx] by[2*oh] ky[oh*ksy] ntx[ow] nty[oh]
struct Tables {
    const int *bx, *kx, *by, *ky, *ntx, *nty;
};
__host__ __device__ inline Tables table_ptrs(const int* t, int oh, int ow, int ksx, int ksy) {
    Tables r;
    r.bx = t; r.kx = r.bx + 2 * ow; r.by = r.kx + (long)ow * ksx; r.ky = r.by + 2 * oh;
    r.ntx = r.ky + (long)oh * ksy; r.nty = r.ntx + ow;
    return r;
}

// KSX / KSY: compile-time tap counts (table rows are zero-padded to them); 0 = generic (runtime count).
// Staging uses aligned dword loads with a per-row byte shift; the horizontal pass keeps a column's weights in
// registers while it walks the rows; the vertical pass walks output rows with wave-uniform (scalar) weights.
template <int KSX, int KSY>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const uint8_t* __restrict__ frames, size_t total_bytes, int Hin, int Win,
                                                             int crop_top, const int* __restrict__ tables, int oh, int ow,
                                                             int ksx, int ksy, int TILE_H, int TILE_W, int max_rows, int in_stride,
                                                             const uint8_t* __restrict__ flip, float* __restrict__ out,
                                                             const int64_t* __restrict__ sel, long pool_frames, int* __restrict__ bad_index) {
    extern __shared__ uint8_t smem[];
    const int h_stride = TILE_W * 3;
    int* s_shift = reinterpret_cast<int*>(smem);                                  // [max_rows + 1] byte shift of each staged row
    uint8_t* s_in = smem + ((size_t)(max_rows + 1) * sizeof(int) + 15) / 16 * 16;   // [max_rows + 1][in_stride]
    uint8_t* s_h = s_in + (size_t)(max_rows + 1) * in_stride;                     // [max_rows + ksy][h_stride]
    const Tables T = table_ptrs(tables, oh, ow, ksx, ksy);
    const int n = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ty0 = blockIdx.y * TILE_H, tx0 = blockIdx.x * TILE_W;
    const int th = min(TILE_H, oh - ty0), tw = min(TILE_W, ow - tx0);
    const int r0 = T.by[2 * ty0], r1 = T.by[2 * (ty0 + th - 1)] + T.by[2 * (ty0 + th - 1) + 1];
    const int c0 = T.bx[2 * tx0], c1 = T.bx[2 * (tx0 + tw - 1)] + T.bx[2 * (tx0 + tw - 1) + 1];
    const int nr = r1 - r0, ncb = (c1 - c0) * 3;
    // sel: batch element n reads frame sel[n] of a resident pool.  An index outside the pool (the reference's tensor indexing raises
    // IndexError) is COUNTED -- the host raises from the count, lf_pipeline.py -- and reads frame 0 here instead of someone else's memory.
    long frame = sel ? sel[n] : n;
    if (frame < 0 || frame >= pool_frames) {
        if (bad_index && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(bad_index, 1);
        frame = 0;
    }
    for (int r = wave; r < nr; r += 4) {
        // sel: batch element n reads frame sel[n] of a resident pool (the gather of a cached dataset's index batch happens here,
        // not in a separate 88 MB copy)
        const size_t g0 = (((size_t)frame * Hin + crop_top + r0 + r) * Win + c0) * 3;
        const int sh = (int)(g0 & 3);
        const size_t a0 = g0 - sh;
        const int nd = (sh + ncb + 3) >> 2;
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_in + (size_t)r * in_stride);
        for (int d = lane; d < nd; d += 64) {
            const size_t byte = a0 + 4 * (size_t)d;
            uint32_t v;
            if (byte + 4 <= total_bytes) {
                v = *reinterpret_cast<const uint32_t*>(frames + byte);
            } else {
                v = 0;
                for (int e = 0; e < 4; ++e)
                    if (byte + e < total_bytes) v |= (uint32_t)frames[byte + e] << (8 * e);
            }
            dst[d] = v;
        }
        if (lane == 0) s_shift[r] = sh;
    }
    __syncthreads();
    // horizontal pass: one (x, channel) column per thread, rows in sequence
    if ((int)threadIdx.x < tw * 3) {
        const int j = threadIdx.x, x = j / 3, c = j - x * 3;
        const int first = (T.bx[2 * (tx0 + x)] - c0) * 3 + c;
        const int* k = T.kx + (long)(tx0 + x) * ksx;
        if (KSX > 0) {
            int w[KSX > 0 ? KSX : 1];
#pragma unroll
            for (int t = 0; t < KSX; ++t) w[t] = k[t];
            for (int r = 0; r < nr; ++r) {
                const uint8_t* p = s_in + (size_t)r * in_stride + s_shift[r] + first;
                int acc = 1 << (PIL_PRECISION_BITS - 1);
#pragma unroll
                for (int t = 0; t < KSX; ++t) acc += (int)p[t * 3] * w[t];     // taps past the count have zero weight
                s_h[r * h_stride + j] = (uint8_t)clip8(acc);
            }
        } else {
            const int cnt = T.bx[2 * (tx0 + x) + 1];
            for (int r = 0; r < nr; ++r) {
                const uint8_t* p = s_in + (size_t)r * in_stride + s_shift[r] + first;
                int acc = 1 << (PIL_PRECISION_BITS - 1);
                for (int t = 0; t < cnt; ++t) acc += (int)p[t * 3] * k[t];
                s_h[r * h_stride + j] = (uint8_t)clip8(acc);
            }
        }
    }
    __syncthreads();
    // vertical pass + ToTensor: fp32 = uint8 / 255, NCHW, optional horizontal flip; thread = (channel, x), x fastest
    const bool fl = flip && flip[n];
    for (int j = threadIdx.x; j < 3 * tw; j += 256) {
        const int c = j / tw, x = j - c * tw;
        const int xo = fl ? ow - 1 - (tx0 + x) : tx0 + x;
        float* o = out + (((size_t)n * 3 + c) * oh + ty0) * ow + xo;
        const uint8_t* col = s_h + x * 3 + c;
        for (int y = 0; y < th; ++y) {
            const int first = T.by[2 * (ty0 + y)] - r0;              // wave-uniform: scalar loads
            const int* k = T.ky + (long)(ty0 + y) * ksy;
            const uint8_t* p = col + first * h_stride;
            int acc = 1 << (PIL_PRECISION_BITS - 1);
            if (KSY > 0) {
#pragma unroll
                for (int t = 0; t < KSY; ++t) acc += (int)p[t * h_stride] * k[t];
            } else {
                const int cnt = T.by[2 * (ty0 + y) + 1];
                for (int t = 0; t < cnt; ++t) acc += (int)p[t * h_stride] * k[t];
            }
            o[(size_t)y * ow] = (float)clip8(acc) / 255.0f;
        }
    }
}

// mode bits: 1 = zero classes 3 and 4 before flipping (BEV always; BP when nclasses < 3),
//            2 = BP flip quirk: positions whose UNFLIPPED label was 3 / 4 become 4 / 3 after the flip
__global__ __launch_bounds__(256) void label_kernel(const uint8_t* __restrict__ labels, int Hin, int Win, int crop_top,
                                                   const int* __restrict__ tables, int oh, int ow, int ksx, int ksy,
                                                   const uint8_t* __restrict__ flip, int mode,
                                                   const int64_t* __restrict__ lut, int64_t* __restrict__ out,
                                                   const int64_t* __restrict__ sel, long pool_frames, int* __restrict__ bad_index) {
    const Tables T = table_ptrs(tables, oh, ow, ksx, ksy);
    const int n = blockIdx.z, y = blockIdx.y;
    long frame = sel ? sel[n] : n;
    if (frame < 0 || frame >= pool_frames) {          // counted for the host (IndexError), never dereferenced
        if (bad_index && blockIdx.x == 0 && y == 0 && threadIdx.x == 0) atomicAdd(bad_index, 1);
        frame = 0;
    }
    const uint8_t* row = labels + ((size_t)frame * Hin + crop_top + T.nty[y]) * Win;
    const bool fl = flip && flip[n];
    for (int x = blockIdx.x * 256 + threadIdx.x; x < ow; x += gridDim.x * 256) {
        const int u = row[T.ntx[x]];                      // unflipped label at (y, x)
        int v;
        if (!fl) {
            v = ((mode & 1) && (u == 3 || u == 4)) ? 0 : u;
        } else {
            int w = row[T.ntx[ow - 1 - x]];
            if ((mode & 1) && (w == 3 || w == 4)) w = 0;
            v = w == 1 ? 2 : (w == 2 ? 1 : w);
            if (mode & 2) {
                if (u == 3) v = 4;
                else if (u == 4) v = 3;
            }
        }
        out[((size_t)n * oh + y) * ow + x] = lut[v];
    }
}

// BEV horizon target: ones above the first row that holds a label (Load_Data_new.py:103-105); one block per image
__global__ __launch_bounds__(256) void horizon_kernel(const int64_t* __restrict__ gt, int oh, int ow, float* __restrict__ hz) {
    const int n = blockIdx.x;
    const int64_t* g = gt + (size_t)n * oh * ow;
    int y_val = oh;
    for (int y = 0; y < oh; ++y) {
        int any = 0;
        for (int x = threadIdx.x; x < ow; x += 256) any |= g[(size_t)y * ow + x] != 0;
        if (__syncthreads_or(any)) { y_val = y; break; }
    }
    for (int y = threadIdx.x; y < oh; y += 256) hz[(size_t)n * oh + y] = y < y_val ? 1.f : 0.f;
}

}  // namespace

static size_t lf_pipeline_lds_bytes(lf_pipeline_plan* P) {
    // staged rows hold [shift <= 3][window][over-read of up to ksx taps], rounded to dwords; one spare row each
    P->in_stride = (P->max_cols * 3 + 3 + P->ksx * 3 + 3) / 4 * 4;
    return ((size_t)(P->max_rows + 1) * sizeof(int) + 15) / 16 * 16 + (size_t)(P->max_rows + 1) * P->in_stride +
           (size_t)(P->max_rows + P->ksy) * P->tile_w * 3;
}

extern "C" {

// Plan for frames of Hin x Win whose bottom `crop_h` rows (starting at crop_top) are resized to out_h x out_w.
lf_pipeline_plan* lf_pipeline_plan_create(int Hin, int Win, int crop_top, int crop_h, int out_h, int out_w) {
    if (Hin < 1 || Win < 1 || crop_top < 0 || crop_h < 1 || crop_top + crop_h > Hin || out_h < 1 || out_w < 1) {
        lf_fail("lf_pipeline_plan_create: bad geometry Hin=%d Win=%d crop=%d+%d out=%dx%d", Hin, Win, crop_top, crop_h, out_h, out_w);
        return nullptr;
    }
    lf_pipeline_plan* P = new lf_pipeline_plan();
    P->Hin = Hin; P->Win = Win; P->crop_top = crop_top; P->crop_h = crop_h; P->out_h = out_h; P->out_w = out_w;
    P->ksx = pil_bilinear_coeffs(Win, out_w, P->bx, P->kx);
    P->ksy = pil_bilinear_coeffs(crop_h, out_h, P->by, P->ky);
    pil_nearest_table(Win, out_w, P->ntx);
    pil_nearest_table(crop_h, out_h, P->nty);
    P->table_ints = (long)(P->bx.size() + P->kx.size() + P->by.size() + P->ky.size() + P->ntx.size() + P->nty.size());
    P->tile_h = TILE_H_MAX; P->tile_w = TILE_W_MAX;
    for (;;) {
        P->max_rows = P->max_cols = 0;
        for (int y0 = 0; y0 < out_h; y0 += P->tile_h) {
            const int y1 = (y0 + P->tile_h < out_h ? y0 + P->tile_h : out_h) - 1;
            const int n = P->by[2 * y1] + P->by[2 * y1 + 1] - P->by[2 * y0];
            if (n > P->max_rows) P->max_rows = n;
        }
        for (int x0 = 0; x0 < out_w; x0 += P->tile_w) {
            const int x1 = (x0 + P->tile_w < out_w ? x0 + P->tile_w : out_w) - 1;
            const int n = P->bx[2 * x1] + P->bx[2 * x1 + 1] - P->bx[2 * x0];
            if (n > P->max_cols) P->max_cols = n;
        }
        P->lds_bytes = lf_pipeline_lds_bytes(P);
        if (P->lds_bytes <= LDS_BUDGET) return P;
        if (P->tile_h == 1 && P->tile_w == 1) break;
        if (P->max_cols >= 2 * P->max_rows && P->tile_w > 1) P->tile_w /= 2;     // keep rows of >= 16 pixels while possible
        else if (P->tile_h > 1) P->tile_h /= 2;
        else P->tile_w /= 2;
    }
    lf_fail("lf_pipeline_plan_create: scale factor too large for the LDS tile");
    delete P;
    return nullptr;
}
void lf_pipeline_plan_destroy(lf_pipeline_plan* P) { delete P; }
size_t lf_pipeline_table_bytes(const lf_pipeline_plan* P) { return (size_t)P->table_ints * sizeof(int); }

// Copy the tables (bilinear bounds + weights, nearest indices) to a device buffer of lf_pipeline_table_bytes().
int lf_pipeline_upload(const lf_pipeline_plan* P, void* tables_dev, void* stream) {
    LF_REQUIRE(P && tables_dev, "lf_pipeline_upload: null pointer");
    std::vector<int> all;
    all.reserve(P->table_ints);
    for (const std::vector<int>* v : {&P->bx, &P->kx, &P->by, &P->ky, &P->ntx, &P->nty}) all.insert(all.end(), v->begin(), v->end());
    // pageable source: the copy is complete when hipMemcpyAsync returns, `all` may go out of scope
    if (hipMemcpyAsync(tables_dev, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess)
        return lf_fail("lf_pipeline_upload: copy failed");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return lf_fail("lf_pipeline_upload: sync failed");
    return 0;
}

// Host copies of the tables for tests (any pointer may be NULL).  Sizes: bounds 2*out, weights out*ksize.
int lf_pipeline_tables_host(const lf_pipeline_plan* P, int* ksx, int* ksy, int* bx, int* kx, int* by, int* ky, int* ntx, int* nty) {
    LF_REQUIRE(P, "lf_pipeline_tables_host: null plan");
    if (ksx) *ksx = P->ksx;
    if (ksy) *ksy = P->ksy;
    if (bx) memcpy(bx, P->bx.data(), P->bx.size() * sizeof(int));
    if (kx) memcpy(kx, P->kx.data(), P->kx.size() * sizeof(int));
    if (by) memcpy(by, P->by.data(), P->by.size() * sizeof(int));
    if (ky) memcpy(ky, P->ky.data(), P->ky.size() * sizeof(int));
    if (ntx) memcpy(ntx, P->ntx.data(), P->ntx.size() * sizeof(int));
    if (nty) memcpy(nty, P->nty.data(), P->nty.size() * sizeof(int));
    return 0;
}

// frames (N, Hin, Win, 3) uint8 HWC (what the image decoder produces); flip (N) uint8 or NULL;
// out (N, 3, out_h, out_w) fp32 in [0, 1].
static int pipeline_image(const lf_pipeline_plan* P, const uint8_t* frames, long pool_frames, const int64_t* sel, int N,
                          const void* tables_dev, const uint8_t* flip, float* out, int* bad_index, void* stream) {
    LF_REQUIRE(P && frames && tables_dev && out && N > 0 && pool_frames > 0, "lf_pipeline_image: bad arguments");
    const dim3 grid(lf_cdiv(P->out_w, P->tile_w), lf_cdiv(P->out_h, P->tile_h), N);
    const size_t total = (size_t)pool_frames * P->Hin * P->Win * 3;
#define LF_RESIZE_LAUNCH(KX, KY)                                                                                              \
    hipLaunchKernelGGL((resize_bilinear_kernel<KX, KY>), grid, dim3(256), P->lds_bytes, (hipStream_t)stream, frames, total,   \
                       P->Hin, P->Win, P->crop_top, (const int*)tables_dev, P->out_h, P->out_w, P->ksx, P->ksy, P->tile_h,    \
                       P->tile_w, P->max_rows, P->in_stride, flip, out, sel, pool_frames, bad_index)
    if (P->ksx == 7 && P->ksy == 7) LF_RESIZE_LAUNCH(7, 7);          // 640x1280 -> 256x512 (x2.5)
    else if (P->ksx == 5 && P->ksy == 5) LF_RESIZE_LAUNCH(5, 5);     // -> 320x640 (x2), 512x1024 (x1.25)
    else LF_RESIZE_LAUNCH(0, 0);
#undef LF_RESIZE_LAUNCH
    LF_CHECK_LAUNCH("lf_pipeline_image");
    return 0;
}
static int pipeline_label(const lf_pipeline_plan* P, const uint8_t* labels, long pool_frames, const int64_t* sel, int N, const void* tables_dev,
                          const uint8_t* flip, int mode, const int64_t* lut, int64_t* out, float* horizon, int* bad_index, void* stream) {
    LF_REQUIRE(P && labels && tables_dev && lut && out && N > 0 && pool_frames > 0, "lf_pipeline_label: bad arguments");
    const dim3 grid(lf_cdiv(P->out_w, 256), P->out_h, N);
    hipLaunchKernelGGL(label_kernel, grid, dim3(256), 0, (hipStream_t)stream, labels, P->Hin, P->Win, P->crop_top,
                       (const int*)tables_dev, P->out_h, P->out_w, P->ksx, P->ksy, flip, mode, lut, out, sel, pool_frames, bad_index);
    if (horizon)
        hipLaunchKernelGGL(horizon_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, out, P->out_h, P->out_w, horizon);
    LF_CHECK_LAUNCH("lf_pipeline_label");
    return 0;
}

// frames (N, Hin, Win, 3) uint8 HWC (what the image decoder produces); flip (N) uint8 or NULL;
// out (N, 3, out_h, out_w) fp32 in [0, 1].
int lf_pipeline_image(const lf_pipeline_plan* P, const uint8_t* frames, int N, const void* tables_dev, const uint8_t* flip,
                      float* out, void* stream) {
    return pipeline_image(P, frames, N, nullptr, N, tables_dev, flip, out, nullptr, stream);
}
// The same from a RESIDENT POOL of decoded frames (pool_frames, Hin, Win, 3): batch element n is frame sel[n] (int64 on the
// device) -- the index batch of a cached dataset is gathered inside the resize kernel instead of by a copy of N frames first.
// bad_index (device int, may be NULL): incremented once per batch element whose sel[n] lies outside [0, pool_frames); such an
// element reads pool entry 0 (never out of bounds) and the caller raises from the count (the reference's tensor indexing raises
// IndexError).
int lf_pipeline_image_indexed(const lf_pipeline_plan* P, const uint8_t* pool, long pool_frames, const int64_t* sel, int N,
                              const void* tables_dev, const uint8_t* flip, float* out, int* bad_index, void* stream) {
    LF_REQUIRE(sel, "lf_pipeline_image_indexed: null index");
    return pipeline_image(P, pool, pool_frames, sel, N, tables_dev, flip, out, bad_index, stream);
}

// labels (N, Hin, Win) uint8 palette indices; lut (256) int64 = (ToTensor(v) * 255).long(); out (N, 1, out_h, out_w)
// int64; horizon (N, out_h) fp32 or NULL.  mode: see label_kernel.
int lf_pipeline_label(const lf_pipeline_plan* P, const uint8_t* labels, int N, const void* tables_dev, const uint8_t* flip,
                      int mode, const int64_t* lut, int64_t* out, float* horizon, void* stream) {
    return pipeline_label(P, labels, N, nullptr, N, tables_dev, flip, mode, lut, out, horizon, nullptr, stream);
}
int lf_pipeline_label_indexed(const lf_pipeline_plan* P, const uint8_t* pool, long pool_frames, const int64_t* sel, int N,
                              const void* tables_dev, const uint8_t* flip, int mode, const int64_t* lut, int64_t* out, float* horizon,
                              int* bad_index, void* stream) {
    LF_REQUIRE(sel, "lf_pipeline_label_indexed: null index");
    return pipeline_label(P, pool, pool_frames, sel, N, tables_dev, flip, mode, lut, out, horizon, bad_index, stream);
}

}  // extern "C"
