// layout_probe.hip -- gfx950 hardware-layout assumptions of fa_fwd_kernel*.hpp,
// checked on the device (the build container has no GPU).  Each check prints
// PASS/FAIL; on FAIL it dumps enough to re-derive the mapping.
//   1. v_mfma_f32_32x32x16_bf16 A/B/C lane maps   2. v_mfma_f32_16x16x32_bf16
//   3. ds_read_b64_tr_b16 (which lane's address feeds which result element)
//   4. v_permlane32_swap                           5. global_load_lds_dwordx4 lane order
// Build: hipcc --offload-arch=gfx950 -O2 layout_probe.hip -o layout_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__host__ __device__ inline int fA(int i, int k) { return (i * 3 + k * 5) % 7 - 3; }
__host__ __device__ inline int fB(int k, int j) { return (k * 2 + j * 7) % 5 - 2; }

__global__ void k_mfma32(float *out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (lane >> 5) + j;
        a[j] = (__bf16)(float)fA(lane & 31, k);   // A[i = lane&31][k]
        b[j] = (__bf16)(float)fB(k, lane & 31);   // B[k][n = lane&31]
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}

__global__ void k_mfma16(float *out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * (lane >> 4) + j;
        a[j] = (__bf16)(float)fA(lane & 15, k);
        b[j] = (__bf16)(float)fB(k, lane & 15);
    }
    f32x4 c = {0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

// LDS holds 16-bit value = its own element index; every lane reads through an
// address given by the host (bytes), result 4 x u16 per lane.
__global__ void k_trread(const int *addr_bytes, unsigned short *out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4 *)((char *)lds + addr_bytes[lane]));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}

__global__ void k_permlane(unsigned *out) {
    const unsigned lane = threadIdx.x & 63;
    auto r = __builtin_amdgcn_permlane32_swap(lane, lane + 100, false, false);
    out[lane * 2 + 0] = r[0];
    out[lane * 2 + 1] = r[1];
}

__global__ void k_glds(const unsigned *src, unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
    const int lane = threadIdx.x & 63;
    // lane L fetches global chunk perm(L) = (L * 5 + 3) & 63 into the lane-linear LDS slot
    const unsigned *g = src + ((lane * 5 + 3) & 63) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = lds[256 + lane * 4 + j];
}

int main() {
    int fails = 0;
    float *d_f; unsigned short *d_u16; int *d_addr; unsigned *d_u32, *d_src;
    CHECK(hipMalloc(&d_f, 64 * 16 * 4));
    CHECK(hipMalloc(&d_u16, 64 * 4 * 2));
    CHECK(hipMalloc(&d_addr, 64 * 4));
    CHECK(hipMalloc(&d_u32, 64 * 4 * 4));
    CHECK(hipMalloc(&d_src, 64 * 4 * 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s arch %s CUs %d clock %d kHz\n", prop.name, prop.gcnArchName,
           prop.multiProcessorCount, prop.clockRate);

    {   // 1. 32x32x16
        std::vector<float> h(64 * 16);
        k_mfma32<<<1, 64>>>(d_f);
        CHECK(hipMemcpy(h.data(), d_f, 64 * 16 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {
                const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float e = 0;
                for (int k = 0; k < 16; ++k) e += fA(row, k) * fB(k, col);
                if (h[lane * 16 + r] != e) ++bad;
            }
        printf("[1] mfma_f32_32x32x16_bf16 A[i=l&31][8*(l>>5)+j] B[8*(l>>5)+j][n=l&31] C col=l&31 row=(r&3)+8(r>>2)+4(l>>5): %s (%d bad)\n",
               bad ? "FAIL" : "PASS", bad);
        if (bad) { for (int r = 0; r < 16; ++r) printf("  lane0 r%d=%g lane33 r%d=%g\n", r, h[r], r, h[33 * 16 + r]); ++fails; }
    }
    {   // 2. 16x16x32
        std::vector<float> h(64 * 4);
        k_mfma16<<<1, 64>>>(d_f);
        CHECK(hipMemcpy(h.data(), d_f, 64 * 4 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int col = lane & 15, row = 4 * (lane >> 4) + r;
                float e = 0;
                for (int k = 0; k < 32; ++k) e += fA(row, k) * fB(k, col);
                if (h[lane * 4 + r] != e) ++bad;
            }
        printf("[2] mfma_f32_16x16x32_bf16 C col=l&15 row=4(l>>4)+r: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) ++fails;
    }
    {   // 3. transpose read, with the V^T operand addressing of fa_fwd_kernel.hpp
        int addr[64];
        for (int l = 0; l < 64; ++l) {
            const int li = l & 15, lg = l >> 4;
            addr[l] = (4 * (lg >> 1) + (li >> 2)) * 64 + (lg & 1) * 32 + (li & 3) * 8 + 512;
        }
        std::vector<unsigned short> h(64 * 4);
        CHECK(hipMemcpy(d_addr, addr, sizeof(addr), hipMemcpyHostToDevice));
        k_trread<<<1, 64>>>(d_addr, d_u16);
        CHECK(hipMemcpy(h.data(), d_u16, 64 * 4 * 2, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                // model: result[l][j] = u16 at (address of lane (group base + 4j + (l&15)/4)) + (l&3)
                const int src_lane = (l & ~15) + 4 * j + ((l & 15) >> 2);
                const int e = addr[src_lane] / 2 + (l & 3);
                if (h[l * 4 + j] != e) ++bad;
            }
        printf("[3] ds_read_b64_tr_b16 result[l][j] = mem[addr(lane 16*(l>>4) + 4j + (l&15)/4)] + (l&3): %s (%d bad)\n",
               bad ? "FAIL" : "PASS", bad);
        if (bad) {
            ++fails;
            for (int l = 0; l < 64; l += 1)
                printf("  lane %2d addr %4d -> %5d %5d %5d %5d\n", l, addr[l] / 2, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        }
    }
    {   // 4. permlane32_swap(vdst = lane, src = lane + 100)
        std::vector<unsigned> h(128);
        k_permlane<<<1, 64>>>(d_u32);
        CHECK(hipMemcpy(h.data(), d_u32, 128 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (unsigned l = 0; l < 64; ++l) {
            // expected: r0 (new vdst): lanes <32 keep vdst (l), lanes >=32 get src[l-32] (l-32+100)
            //           r1 (new src) : lanes <32 get vdst[l+32] (l+32), lanes >=32 keep src (l+100)
            const unsigned e0 = l < 32 ? l : l - 32 + 100, e1 = l < 32 ? l + 32 : l + 100;
            if (h[l * 2] != e0 || h[l * 2 + 1] != e1) ++bad;
        }
        printf("[4] permlane32_swap half exchange: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) { ++fails; for (int l = 0; l < 64; l += 8) printf("  lane %d -> %u %u\n", l, h[l * 2], h[l * 2 + 1]); }
    }
    {   // 5. global_load_lds: LDS slot L (16 B) <- what lane L addressed
        std::vector<unsigned> src(256), h(256);
        for (int i = 0; i < 256; ++i) src[i] = 1000 * (i / 4) + (i & 3);
        CHECK(hipMemcpy(d_src, src.data(), 1024, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(d_src, d_u32);
        CHECK(hipMemcpy(h.data(), d_u32, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j)
                if (h[l * 4 + j] != 1000u * ((l * 5 + 3) & 63) + j) ++bad;
        printf("[5] global_load_lds_dwordx4 lane-linear destination: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) { ++fails; for (int l = 0; l < 8; ++l) printf("  slot %d -> %u %u %u %u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]); }
    }
    printf("layout_probe: %d failing checks\n", fails);
    return fails ? 1 : 0;
}
