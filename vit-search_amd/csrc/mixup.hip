// SwitchTokenMix on the device (reference token_mixup.py:39-162): the input side of the training step.
// The random draws (two torch.randperm, the Beta samples and the box from numpy) stay on the host with the reference's
// generators and call order (vitres/token_mixup.py); what is heavy -- rewriting the 77 MB image batch and building the
// 8 MB of soft patch targets -- is two streaming kernels, one read and one write per element, bit-exact with the
// reference's fp32 arithmetic (products rounded separately, then added: no FMA contraction).
//   rows [0, half)  : out = partner's pixels inside the box, own pixels outside; patch targets follow the box
//   rows [half, B)  : out = x * lam + partner * (1 - lam); every patch target = the mixed image target
#include "common.h"
#include "../../include/vitres_hip.h"

namespace {

// fl(fl(a * la) + fl(b * lb)): hipcc contracts a*b+c into an FMA by default (-ffp-contract=fast), also through __fmul_rn
__device__ __forceinline__ float mix2(float a, float la, float b, float lb) {
#pragma clang fp contract(off)
    const float p = a * la;
    const float q = b * lb;
    return p + q;
}

__global__ __launch_bounds__(256) void mix_samples_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const int64_t* __restrict__ partner, int B, int C, int H, int W,
                                                          int half, int py0, int py1, int px0, int px1, float lam, float oml) {
    const int w4 = W / 4;
    const long long per = (long long)C * H * w4, total = (long long)B * per;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int b = (int)(idx / per);
        const long long r = idx % per;
        const int x = (int)(r % w4) * 4, yrow = (int)((r / w4) % H);
        const long long off = (long long)b * C * H * W + r * 4;
        const long long poff = (long long)partner[b] * C * H * W + r * 4;
        const float4 a = *reinterpret_cast<const float4*>(in + off);
        float4 o = a;
        if (b < half) {
            if (yrow >= py0 && yrow < py1 && x + 3 >= px0 && x < px1) {
                const float4 p = *reinterpret_cast<const float4*>(in + poff);
                if (x + 0 >= px0 && x + 0 < px1) o.x = p.x;
                if (x + 1 >= px0 && x + 1 < px1) o.y = p.y;
                if (x + 2 >= px0 && x + 2 < px1) o.z = p.z;
                if (x + 3 >= px0 && x + 3 < px1) o.w = p.w;
            }
        } else {
            const float4 p = *reinterpret_cast<const float4*>(in + poff);
            o = make_float4(mix2(a.x, lam, p.x, oml), mix2(a.y, lam, p.y, oml), mix2(a.z, lam, p.z, oml), mix2(a.w, lam, p.w, oml));
        }
        *reinterpret_cast<float4*>(out + off) = o;
    }
}

// one workgroup per sample: targets[b, :] and patch_targets[b, p, :] for all p
__global__ __launch_bounds__(256) void mix_targets_kernel(const int64_t* __restrict__ labels, const int64_t* __restrict__ partner,
                                                          float* __restrict__ targets, float* __restrict__ patch_targets, int K,
                                                          int PL, int half, int y0, int y1, int x0, int x1, float lam_p,
                                                          float oml_p, float lam_i, float oml_i, float on, float off) {
    const int b = blockIdx.x;
    const int la = (int)labels[b], lb = (int)labels[partner[b]];
    const bool patch = b < half;
    const float l = patch ? lam_p : lam_i, m = patch ? oml_p : oml_i;
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
        const float ya = c == la ? on : off, yb = c == lb ? on : off;
        const float t = mix2(ya, l, yb, m);
        targets[(long long)b * K + c] = t;
        for (int p = 0; p < PL * PL; ++p) {
            const int i = p / PL, j = p % PL;
            const bool in = i >= y0 && i < y1 && j >= x0 && j < x1;
            patch_targets[((long long)b * PL * PL + p) * K + c] = patch ? (in ? yb : ya) : t;
        }
    }
}

}  // namespace

extern "C" int vr_token_mix(const float* samples, float* out, const int64_t* labels, const int64_t* partner, float* targets,
                            float* patch_targets, int32_t B, int32_t C, int32_t H, int32_t W, int32_t num_classes,
                            int32_t patch_len, int32_t half, int32_t y0, int32_t y1, int32_t x0, int32_t x1, float lam_patch,
                            float oml_patch, float lam_img, float oml_img, float on_value, float off_value, vr_stream_t stream) {
    if (!samples || !out || !labels || !partner || !targets || !patch_targets) return VR_EINVAL;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || num_classes <= 0 || patch_len <= 0 || half < 0 || half > B) return VR_EINVAL;
    if (W % 4 || H % patch_len || W % patch_len || samples == out) return VR_EUNSUPPORTED;
    if (((uintptr_t)samples & 15) || ((uintptr_t)out & 15)) return VR_EALIGN;
    const int psy = H / patch_len, psx = W / patch_len;
    const long long total = (long long)B * C * H * (W / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(mix_samples_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, samples, out, partner, B, C, H,
                       W, half, psy * y0, psy * y1, psx * x0, psx * x1, lam_img, oml_img);
    hipLaunchKernelGGL(mix_targets_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, labels, partner, targets, patch_targets,
                       num_classes, patch_len, half, y0, y1, x0, x1, lam_patch, oml_patch, lam_img, oml_img, on_value, off_value);
    VR_CHECK_LAUNCH();
    return VR_OK;
}
