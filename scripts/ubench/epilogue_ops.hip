// Micro-test for the instructions the epilogue of k_qgram_mm (csrc/qgram_mm.hpp) leans on:
//   1. v_cvt_pk_u8_f32: rounding (nearest even?) and saturation to [0, 255] -- the bound byte is
//      min(255, floor((t + 8) / 9)) computed as cvt_pk_u8(fma(t + 8, 1 / 9, 0.5 / 9 - 0.5)) for every integer t in [-16384, 16384];
//   2. the 4 x 4 byte transpose over a quad of lanes (DPP quad_perm broadcasts + v_perm_b32);
//   3. the 16-lane row reductions by DPP row_shr (min and sum), result in lane 15 of a row;
//   4. the packed "bytes <= 36" count and the packed 16-bit key minimum of a 16-byte chunk.
// Build: hipcc -O3 --offload-arch=gfx950 -o epilogue_ops epilogue_ops.hip ; exit code 0 = every check passed.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__global__ void k_cvt(const float *in, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1u, 0xaabbccddu);
}

__global__ void k_bound(uint32_t *out)          // t = i - 16384
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float t = (float)(i - 16384);
    const float z = __builtin_fmaf(t + 8.0f, 1.0f / 9.0f, 0.5f / 9.0f - 0.5f);
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(z, 0u, 0u);
}

template <int S> __device__ __forceinline__ uint32_t quad_bcast(uint32_t w)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, S * 0x55, 0xf, 0xf, true);          // quad_perm:[S,S,S,S]
}

__global__ void k_transpose(const uint32_t *in, uint32_t *out)
{
    const int lane = threadIdx.x;
    const uint32_t w = in[lane];
    const uint32_t t = lane & 3;
    const uint32_t a0 = quad_bcast<0>(w), a1 = quad_bcast<1>(w), a2 = quad_bcast<2>(w), a3 = quad_bcast<3>(w);
    const uint32_t sel_lo = 0x0c0c0000u | ((4u + t) << 8) | t, sel_hi = 0x00000c0cu | ((4u + t) << 24) | (t << 16);
    out[lane] = __builtin_amdgcn_perm(a1, a0, sel_lo) | __builtin_amdgcn_perm(a3, a2, sel_hi);
}

template <int CTRL> __device__ __forceinline__ uint32_t dpp_min(uint32_t x)
{
    const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, CTRL, 0xf, 0xf, false);
    return y < x ? y : x;
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_add(uint32_t x)
{
    return x + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}

__global__ void k_rowreduce(const uint32_t *in, uint32_t *omin, uint32_t *osum)
{
    const int lane = threadIdx.x;
    uint32_t a = in[lane], b = in[lane];
    a = dpp_min<0x111>(a); a = dpp_min<0x112>(a); a = dpp_min<0x114>(a); a = dpp_min<0x118>(a);          // row_shr:1, 2, 4, 8
    b = dpp_add<0x111>(b); b = dpp_add<0x112>(b); b = dpp_add<0x114>(b); b = dpp_add<0x118>(b);
    omin[lane] = a; osum[lane] = b;
}

__global__ void k_packed(const uint4 *in, uint32_t *ohub, uint32_t *okey, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 v = in[i];
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
    uint32_t hub = 0;
    us2 m = {0xffff, 0xffff};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const uint32_t w = wv[d];
        const uint32_t gt = (((w & 0x7f7f7f7fu) + 0x5b5b5b5bu) | w) & 0x80808080u;          // 0x80 per byte > 36
        hub += (uint32_t)__builtin_popcount(gt ^ 0x80808080u);
        const uint32_t ce = (uint32_t)(4 * d) | ((uint32_t)(4 * d + 2) << 16), co = (uint32_t)(4 * d + 1) | ((uint32_t)(4 * d + 3) << 16);
        const uint32_t ke = ((w << 8) & 0xff00ff00u) | ce, ko = (w & 0xff00ff00u) | co;
        us2 e, o;
        __builtin_memcpy(&e, &ke, 4); __builtin_memcpy(&o, &ko, 4);
        m = __builtin_elementwise_min(m, __builtin_elementwise_min(e, o));
    }
    ohub[i] = hub;
    okey[i] = m.x < m.y ? m.x : m.y;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main()
{
    int bad = 0;
    {   // 1a. rounding and saturation
        const float vals[] = {-5.f, -0.6f, -0.5f, -0.4f, 0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.4f, 255.5f, 256.f, 300.f, 1e9f};
        const int n = sizeof(vals) / sizeof(float);
        float *d_in; uint32_t *d_out;
        CK(hipMalloc(&d_in, n * 4)); CK(hipMalloc(&d_out, n * 4));
        CK(hipMemcpy(d_in, vals, n * 4, hipMemcpyHostToDevice));
        k_cvt<<<1, 64>>>(d_in, d_out, n);
        std::vector<uint32_t> o(n);
        CK(hipMemcpy(o.data(), d_out, n * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g, byte 1, 0xaabbccdd) = 0x%08x\n", vals[i], o[i]);
    }
    {   // 1b. the bound formula over the whole range
        uint32_t *d; CK(hipMalloc(&d, 32769 * 4));
        k_bound<<<(32769 + 255) / 256, 256>>>(d);
        std::vector<uint32_t> o(32769 + 255);
        CK(hipMemcpy(o.data(), d, 32769 * 4, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int i = 0; i < 32769; ++i) {
            const int t = i - 16384;
            const int v = t < 0 ? 0 : t;
            int want = (v + 8) / 9; if (want > 255) want = 255;
            if ((int)o[i] != want) { if (wrong < 5) printf("bound(t=%d) = %u, want %d\n", t, o[i], want); ++wrong; }
        }
        printf("bound formula: %d of 32769 wrong\n", wrong);
        bad += wrong != 0;
    }
    {   // 2. quad transpose
        std::vector<uint32_t> in(64), o(64);
        for (int l = 0; l < 64; ++l) in[l] = 0x01000000u * (4 * l + 3) | 0x010000u * (4 * l + 2) | 0x0100u * (4 * l + 1) | (uint32_t)(4 * l);
        uint32_t *d_in, *d_out; CK(hipMalloc(&d_in, 256)); CK(hipMalloc(&d_out, 256));
        CK(hipMemcpy(d_in, in.data(), 256, hipMemcpyHostToDevice));
        k_transpose<<<1, 64>>>(d_in, d_out);
        CK(hipMemcpy(o.data(), d_out, 256, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int l = 0; l < 64; ++l) {
            const int q = l & ~3, t = l & 3;
            uint32_t want = 0;
            for (int s = 0; s < 4; ++s) want |= ((in[q + s] >> (8 * t)) & 0xffu) << (8 * s);
            if (o[l] != want) { if (wrong < 5) printf("transpose lane %d: 0x%08x want 0x%08x\n", l, o[l], want); ++wrong; }
        }
        printf("quad transpose: %d of 64 wrong\n", wrong);
        bad += wrong != 0;
    }
    {   // 3. row reductions
        std::vector<uint32_t> in(64), mn(64), sm(64);
        srand(7);
        for (int l = 0; l < 64; ++l) in[l] = (uint32_t)rand() % 100000u;
        uint32_t *d_in, *d_a, *d_b; CK(hipMalloc(&d_in, 256)); CK(hipMalloc(&d_a, 256)); CK(hipMalloc(&d_b, 256));
        CK(hipMemcpy(d_in, in.data(), 256, hipMemcpyHostToDevice));
        k_rowreduce<<<1, 64>>>(d_in, d_a, d_b);
        CK(hipMemcpy(mn.data(), d_a, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(sm.data(), d_b, 256, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int row = 0; row < 4; ++row) {
            uint32_t wmin = ~0u, wsum = 0;
            for (int l = 0; l < 16; ++l) { wmin = in[row * 16 + l] < wmin ? in[row * 16 + l] : wmin; wsum += in[row * 16 + l]; }
            if (mn[row * 16 + 15] != wmin || sm[row * 16 + 15] != wsum) { printf("row %d: min %u want %u, sum %u want %u\n", row, mn[row * 16 + 15], wmin, sm[row * 16 + 15], wsum); ++wrong; }
        }
        printf("row reductions: %d of 4 wrong\n", wrong);
        bad += wrong != 0;
    }
    {   // 4. packed count and minimum
        const int n = 4096;
        std::vector<uint4> in(n);
        std::vector<uint32_t> hub(n), key(n);
        srand(11);
        for (int i = 0; i < n; ++i) {
            uint32_t w[4];
            for (int d = 0; d < 4; ++d) { w[d] = 0; for (int k = 0; k < 4; ++k) { const int mode = rand() % 4; const uint32_t b = mode == 0 ? rand() % 256 : mode == 1 ? 30 + rand() % 14 : mode == 2 ? 120 + rand() % 16 : rand() % 64; w[d] |= b << (8 * k); } }
            in[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        uint4 *d_in; uint32_t *d_h, *d_k; CK(hipMalloc(&d_in, n * 16)); CK(hipMalloc(&d_h, n * 4)); CK(hipMalloc(&d_k, n * 4));
        CK(hipMemcpy(d_in, in.data(), n * 16, hipMemcpyHostToDevice));
        k_packed<<<n / 256, 256>>>(d_in, d_h, d_k, n);
        CK(hipMemcpy(hub.data(), d_h, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(key.data(), d_k, n * 4, hipMemcpyDeviceToHost));
        int wrong = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t w[4] = {in[i].x, in[i].y, in[i].z, in[i].w};
            uint32_t wh = 0, wk = ~0u;
            for (int k = 0; k < 16; ++k) { const uint32_t b = (w[k >> 2] >> (8 * (k & 3))) & 0xffu; wh += b <= 36; const uint32_t c = (b << 8) | (uint32_t)k; wk = c < wk ? c : wk; }
            if (hub[i] != wh || key[i] != wk) { if (wrong < 5) printf("packed %d: hub %u want %u, key 0x%x want 0x%x\n", i, hub[i], wh, key[i], wk); ++wrong; }
        }
        printf("packed count / minimum: %d of %d wrong\n", wrong, n);
        bad += wrong != 0;
    }
    CK(hipDeviceSynchronize());
    printf(bad ? "FAILED\n" : "all checks passed\n");
    return bad ? 1 : 0;
}
