// the training-time frame transform on the device (reference: multimodal_data_module.py:244-256 --
// RandomResizedCrop((224,224), scale (0.2,1)) -> RandomApply([GaussianBlur([.1,2.])], p .5) (utils.py:94-103) ->
// RandomHorizontalFlip -> ToTensor -> Normalize (:57)).  The reference runs it per frame on PIL images inside 8 DataLoader
// workers; here the decoded uint8 frames are resident in HBM and ONE launch transforms a whole batch, bit-identically to
// Pillow's integer pixel arithmetic (oracle/augment_oracle.py, pinned against Pillow itself):
//
//   crop + bilinear resize   Pillow Resample.c: separable triangle filter widened by the down-scale factor; coefficients
//                            normalised in double, rounded to 22 fractional bits; horizontal pass, then vertical pass, each
//                            rounded to uint8
//   Gaussian blur            Pillow BoxBlur.c: three box blurs of a fractional radius along x, then three along y, 24-bit
//                            fixed point, edge pixels replicated
//   flip / ToTensor / Normalize   (u8 / 255 - mean) / std in fp32 (true divisions, as torch), written NCHW
//
// One workgroup per (frame, colour plane): a 224 x 224 plane is 49 KB, so the horizontally resampled crop, the resized plane
// and the blur's ping-pong all live in LDS (<= 160 KB) and the only HBM traffic is the crop read (once per plane) and the
// fp32 plane write -- 3 x 256 = 768 workgroups for the BASELINE batch, 3 per CU.  The random draws (crop box, blur sigma,
// flip) are made by the host (multimodal/augment.py) and passed as small device arrays; no pixel work runs on the CPU.
#include "cvcl_common.h"

namespace {

constexpr int AUG_PB = 22;                 // Pillow's PRECISION_BITS = 32 - 8 - 2

struct AugDev {
    const unsigned char* frames;           // [B][H][W][3]
    const int* crop;                       // [B][4] top, left, h, w
    const float* sigma;                    // [B]  <= 0: no blur
    const int* flip;                       // [B]
    float* out;                            // [B][3][OH][OW]
    unsigned char* out_u8;                 // optional [B][OH][OW][3] (the uint8 image before ToTensor), may be NULL
    int B, H, W, OH, OW;
    int t_rows;                            // rows of the LDS buffer T (>= max crop height, >= OH)
    int kmax_h, kmax_v;                    // coefficient slots per output index of the horizontal / vertical pass
    float mean[3], stdv[3];
};

// Pillow precompute_coeffs + normalize_coeffs_8bpc for output index xx (bilinear): bounds and integer taps into kk[0..kmax)
#pragma clang fp contract(off)
__device__ inline void resample_taps(int in_size, int out_size, int xx, int kmax, int* bounds, int* kk) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * ss;
        t = t < 0.0 ? -t : t;
        ww += t < 1.0 ? 1.0 - t : 0.0;
    }
    for (int x = 0; x < kmax; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double t = (x + xmin - center + 0.5) * ss;
            t = t < 0.0 ? -t : t;
            w = t < 1.0 ? 1.0 - t : 0.0;
            if (ww != 0.0) w /= ww;
        }
        kk[x] = (int)(w * (double)(1 << AUG_PB) + (w < 0.0 ? -0.5 : 0.5));
    }
    bounds[0] = xmin;
    bounds[1] = xmax;
}

// Pillow _gaussian_blur_radius (3 passes) + ImagingHorizontalBoxBlur's fixed-point weights
#pragma clang fp contract(off)
__device__ inline void box_weights(float radius, int* r_out, unsigned* ww_out, unsigned* fw_out) {
    const float sigma2 = radius * radius / 3.0f;
    const float L = (float)sqrt(12.0 * (double)sigma2 + 1.0);
    const float l = (float)floor(((double)L - 1.0) / 2.0);
    float a = (2.0f * l + 1.0f) * (l * (l + 1.0f) - 3.0f * sigma2);
    a = a / (6.0f * (sigma2 - (l + 1.0f) * (l + 1.0f)));
    const float fr = l + a;
    const int r = (int)fr;
    const unsigned ww = (unsigned)(16777216.0f / (fr * 2.0f + 1.0f));
    *r_out = r;
    *ww_out = ww;
    *fw_out = ((1u << 24) - (unsigned)(r * 2 + 1) * ww) / 2u;
}

// one box-blur pass along a line direction: n = line length, lines = number of lines; element (line, x) at
// line * line_stride + x * x_stride
__device__ inline void box_pass(const unsigned char* src, unsigned char* dst, int lines, int n, int line_stride, int x_stride, int r,
                                unsigned ww, unsigned fw) {
    // consecutive threads walk the contiguous direction of the plane whatever the blur axis: i = slow * fast_n + fast with
    // (slow, fast) = (line, x) for the row blur and (x, line) for the column blur, advanced without divisions
    const int fast_n = x_stride == 1 ? n : lines;
    int slow = threadIdx.x / fast_n, fast = threadIdx.x - slow * fast_n;
    const int dslow = blockDim.x / fast_n, dfast = blockDim.x - dslow * fast_n;
    for (int i = threadIdx.x; i < lines * n; i += blockDim.x) {
        const int line = x_stride == 1 ? slow : fast;
        const int x = x_stride == 1 ? fast : slow;
        slow += dslow; fast += dfast;
        if (fast >= fast_n) { fast -= fast_n; ++slow; }
        const unsigned char* ln = src + line * line_stride;
        unsigned acc = 0;
        for (int d = -r; d <= r; ++d) {
            int xi = x + d;
            xi = xi < 0 ? 0 : (xi > n - 1 ? n - 1 : xi);
            acc += ln[xi * x_stride];
        }
        int xl = x - r - 1, xr = x + r + 1;
        xl = xl < 0 ? 0 : xl;
        xr = xr > n - 1 ? n - 1 : xr;
        const unsigned bulk = acc * ww + ((unsigned)ln[xl * x_stride] + (unsigned)ln[xr * x_stride]) * fw;
        dst[line * line_stride + x * x_stride] = (unsigned char)((bulk + (1u << 23)) >> 24);
    }
}

__global__ __launch_bounds__(1024) void augment_frames_kernel(AugDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / 3, c = blockIdx.x % 3;
    const int OH = p.OH, OW = p.OW;
    unsigned char* T = (unsigned char*)smem;                       // [t_rows][OW]: horizontally resampled crop, later blur ping-pong
    unsigned char* P = T + (size_t)p.t_rows * OW;                  // [OH][OW]: the resized plane
    int* kH = (int*)P;                                             // [OW][kmax_h] taps + [OW][2] bounds: dead before P is written
    int* bH = kH + OW * p.kmax_h;
    int* kV = (int*)(P + (((size_t)OH * OW + 15) & ~(size_t)15));  // [OH][kmax_v], [OH][2]
    int* bV = kV + OH * p.kmax_v;

    int top = p.crop[b * 4 + 0], left = p.crop[b * 4 + 1], h = p.crop[b * 4 + 2], w = p.crop[b * 4 + 3];
    // (the host validates the boxes; a malformed box would index outside the frame, so clamp defensively)
    h = h < 1 ? 1 : (h > p.H ? p.H : h);
    w = w < 1 ? 1 : (w > p.W ? p.W : w);
    if (h > p.t_rows) h = p.t_rows;
    top = top < 0 ? 0 : (top > p.H - h ? p.H - h : top);
    left = left < 0 ? 0 : (left > p.W - w ? p.W - w : left);

    for (int i = threadIdx.x; i < OW + OH; i += blockDim.x) {
        if (i < OW) resample_taps(w, OW, i, p.kmax_h, bH + i * 2, kH + i * p.kmax_h);
        else resample_taps(h, OH, i - OW, p.kmax_v, bV + (i - OW) * 2, kV + (i - OW) * p.kmax_v);
    }
    __syncthreads();

    // horizontal pass: crop rows (global, HWC bytes of plane c) -> T[h][OW]
    const unsigned char* src = p.frames + ((size_t)b * p.H + top) * p.W * 3 + (size_t)left * 3 + c;
    const int dy = blockDim.x / OW, dx = blockDim.x - dy * OW;    // (row, column) of element i advanced without divisions
    int y = threadIdx.x / OW, xx = threadIdx.x - y * OW;
    for (int i = threadIdx.x; i < h * OW; i += blockDim.x) {
        const int xmin = bH[xx * 2], xmax = bH[xx * 2 + 1];
        const unsigned char* row = src + (size_t)y * p.W * 3 + (size_t)xmin * 3;
        const int* k = kH + xx * p.kmax_h;
        int ss = 1 << (AUG_PB - 1);
        for (int x = 0; x < xmax; ++x) ss += (int)row[x * 3] * k[x];
        ss >>= AUG_PB;
        T[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
        y += dy; xx += dx;
        if (xx >= OW) { xx -= OW; ++y; }
    }
    __syncthreads();
    // vertical pass: T[h][OW] -> P[OH][OW]
    int yy = threadIdx.x / OW, x = threadIdx.x - yy * OW;
    for (int i = threadIdx.x; i < OH * OW; i += blockDim.x) {
        const int ymin = bV[yy * 2], ymax = bV[yy * 2 + 1];
        const int* k = kV + yy * p.kmax_v;
        int ss = 1 << (AUG_PB - 1);
        for (int y = 0; y < ymax; ++y) ss += (int)T[(ymin + y) * OW + x] * k[y];
        ss >>= AUG_PB;
        P[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
        yy += dy; x += dx;
        if (x >= OW) { x -= OW; ++yy; }
    }
    __syncthreads();

    const float sigma = p.sigma[b];
    if (sigma > 0.f) {                                             // uniform per workgroup
        int r;
        unsigned ww, fw;
        box_weights(sigma, &r, &ww, &fw);
        // three passes along x (lines = rows), then three along y (lines = columns), ping-pong P <-> T
        box_pass(P, T, OH, OW, OW, 1, r, ww, fw); __syncthreads();
        box_pass(T, P, OH, OW, OW, 1, r, ww, fw); __syncthreads();
        box_pass(P, T, OH, OW, OW, 1, r, ww, fw); __syncthreads();
        box_pass(T, P, OW, OH, 1, OW, r, ww, fw); __syncthreads();
        box_pass(P, T, OW, OH, 1, OW, r, ww, fw); __syncthreads();
        box_pass(T, P, OW, OH, 1, OW, r, ww, fw); __syncthreads();
    }

    const bool flip = p.flip[b] != 0;
    const float mean = p.mean[c], stdv = p.stdv[c];
    float* o = p.out + ((size_t)b * 3 + c) * OH * OW;
    int oy = threadIdx.x / OW, ox = threadIdx.x - oy * OW;
    for (int i = threadIdx.x; i < OH * OW; i += blockDim.x) {
        const unsigned char v = P[oy * OW + (flip ? OW - 1 - ox : ox)];
        oy += dy; ox += dx;
        if (ox >= OW) { ox -= OW; ++oy; }
        o[i] = __fdiv_rn(__fdiv_rn((float)v, 255.0f) - mean, stdv);
        if (p.out_u8) p.out_u8[((size_t)b * OH * OW + i) * 3 + c] = v;
    }
}

}  // namespace

static size_t aug_lds_bytes(int t_rows, int OH, int OW, int kmax_v) {
    return (size_t)t_rows * OW + (((size_t)OH * OW + 15) & ~(size_t)15) + (size_t)OH * (kmax_v + 2) * sizeof(int);
}
static int aug_taps(double scale) { return (int)ceil(scale < 1.0 ? 1.0 : scale) * 2 + 1; }   // Pillow: ksize = ceil(support) * 2 + 1

// frames: uint8 [B][H][W][3] (decoded RGB frames, HWC as PIL / the image files hold them); crop: int32 [B][4] = top, left, h, w
// (torchvision RandomResizedCrop.get_params order); blur_sigma: fp32 [B], <= 0 where RandomApply skipped the blur; flip: int32
// [B]; mean / std: 3 host floats each; out: fp32 [B][3][out_h][out_w]; out_u8 (optional): the uint8 image before ToTensor.
// max_crop_h: an upper bound of crop[:, 2] (sizes the LDS plan; H always works).  All arrays but mean / std are device memory.
extern "C" int cvcl_augment_frames(const void* frames, int B, int H, int W, const int32_t* crop, const float* blur_sigma,
                                   const int32_t* flip, const float* mean, const float* std3, void* out, int out_h, int out_w,
                                   void* out_u8, int max_crop_h, void* stream) {
    CVCL_CHECK_ARG(frames && crop && blur_sigma && flip && mean && std3 && out, "cvcl_augment_frames: null operand");
    CVCL_CHECK_ARG(B > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0 && max_crop_h > 0 && max_crop_h <= H,
                   "cvcl_augment_frames: bad sizes (B %d, frame %d x %d, output %d x %d, max crop height %d)", B, H, W, out_h, out_w,
                   max_crop_h);
    AugDev d;
    d.frames = (const unsigned char*)frames; d.crop = crop; d.sigma = blur_sigma; d.flip = flip;
    d.out = (float*)out; d.out_u8 = (unsigned char*)out_u8;
    d.B = B; d.H = H; d.W = W; d.OH = out_h; d.OW = out_w;
    d.t_rows = max_crop_h > out_h ? max_crop_h : out_h;
    d.kmax_h = aug_taps((double)W / out_w);                // upper bounds: the widest / tallest crop
    d.kmax_v = aug_taps((double)max_crop_h / out_h);
    CVCL_CHECK_ARG((size_t)out_w * (d.kmax_h + 2) * sizeof(int) <= (size_t)out_h * out_w,
                   "cvcl_augment_frames: %d-tap horizontal filter table does not fit the plane buffer", d.kmax_h);
    for (int i = 0; i < 3; ++i) { d.mean[i] = mean[i]; d.stdv[i] = std3[i]; }
    const size_t lds = aug_lds_bytes(d.t_rows, out_h, out_w, d.kmax_v);
    CVCL_CHECK_ARG(lds <= 160 * 1024,
                   "cvcl_augment_frames: a %d-row crop resampled to %d x %d needs %zu bytes of LDS (limit 163840): crop boxes that tall "
                   "are not supported by the single-pass plan", max_crop_h, out_h, out_w, lds);
    static CvclLdsAttr attr_set;
    if (!attr_set.ready()) {
        if (hipFuncSetAttribute((const void*)augment_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            cvcl_set_error("cvcl_augment_frames: cannot raise the dynamic LDS limit");
            return CVCL_ELAUNCH;
        }
        attr_set.mark();
    }
    hipLaunchKernelGGL(augment_frames_kernel, dim3(B * 3), dim3(1024), lds, (hipStream_t)stream, d);   // one workgroup per CU (LDS): 16 waves hide the LDS latency
    CVCL_LAUNCH_CHECK();
    return CVCL_OK;
}
