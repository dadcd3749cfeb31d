// What exactly does `buffer_load_dwordx4 v_off, s[rsrc], s_off offen offset:IMM lds` do on gfx950?
//   (1) where does lane l's 16 bytes land: M0 + IMM + 16 l ?            (2) is IMM added to BOTH the memory and the LDS address?
//   (3) does M0 reach LDS addresses above 64 KB / 128 KB (160 KB LDS)?   (4) is the data visible to other waves after the issuer's vmcnt(0) + s_barrier?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_lds_dma.hip -o exp/probe_lds_dma ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) u32x4 smem[];

__global__ __launch_bounds__(256) void k(const unsigned* src, unsigned nbytes, unsigned lds_base, unsigned* out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (unsigned i = threadIdx.x; i < 40960 / 4; i += 256) smem[i] = u32x4{0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
    // wave w copies 2 KB: memory [w * 2048, +2048) -> LDS [lds_base + w * 2048, +2048), second KB through offset:1024
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_base + wv * 2048u), soff = __builtin_amdgcn_readfirstlane(wv * 2048u);
    const unsigned voff = lane * 16u;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
                 :: "s"(m0v), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    // every wave reads the 2 KB of wave (w + 1) % 4
    const unsigned rw = (wv + 1) & 3;
    typedef __attribute__((address_space(3))) u32x4 lds_t;
    const u32x4 a = *reinterpret_cast<const lds_t*>((size_t)(lds_base + rw * 2048u + lane * 16u));
    const u32x4 b = *reinterpret_cast<const lds_t*>((size_t)(lds_base + rw * 2048u + 1024u + lane * 16u));
    unsigned* o = out + (size_t)threadIdx.x * 8;
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

int main()
{
    const unsigned n = 8192 / 4;
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; ++i) h[i] = i;
    unsigned *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 8 * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad_total = 0;
    for (unsigned base : {0u, 1024u, 32768u, 65536u - 4096u, 65536u, 100000u / 16 * 16, 131072u, 160u * 1024u - 8192u}) {
        hipMemset(out, 0xff, 256 * 8 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, src, n * 4, base, out);
        hipDeviceSynchronize();
        std::vector<unsigned> r(256 * 8);
        hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0; unsigned first_bad = 0, first_val = 0;
        for (int t = 0; t < 256; ++t) {
            const int lane = t & 63, wv = t >> 6, rw = (wv + 1) & 3;
            for (int e = 0; e < 8; ++e) {
                const unsigned want = (rw * 2048u + (e >= 4 ? 1024u : 0u) + lane * 16u) / 4 + (e & 3);
                if (r[t * 8 + e] != want) { if (!bad) { first_bad = t * 8 + e; first_val = r[t * 8 + e]; } ++bad; }
            }
        }
        printf("lds_base %6u: %s (%d wrong of 2048; first wrong at %u = 0x%x)\n", base, bad ? "MISMATCH" : "ok", bad, first_bad, first_val);
        bad_total += bad;
    }
    printf(bad_total ? "LDS-DMA probe: FAILED\n" : "LDS-DMA probe: all layouts as assumed (M0 + IMM + 16 * lane, IMM added to both addresses, visible after vmcnt(0) + barrier)\n");
    return bad_total != 0;
}
