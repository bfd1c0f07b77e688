// Does a returning LDS atomic (ds_add_rtn_u32) hand out ranks in ascending LANE order to the lanes of one instruction that
// hit the same address?  Nothing documents the LDS arbiter's order, so this probe measures it: every wavefront draws 64 keys
// from a small range (heavy collisions), does ONE atomicAdd(&cnt[key >> 1], 1 << 16 * (key & 1)) per lane and compares the
// old half-word it got back with the number of LOWER lanes holding the same key.  Many workgroups, many rounds, several key
// ranges, 1 to 16 wavefronts per workgroup hammering separate counter rows (bank conflicts between wavefronts included).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_order_probe.hip -o gpurun_out/lds_probe && ./gpurun_out/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(int range, int rounds, unsigned long long* bad, unsigned long long* total) {
    __shared__ uint32_t cnt[WAVES][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t s = 0x9e3779b9u * (blockIdx.x * 1024u + threadIdx.x + 1u);
    unsigned long long nbad = 0, ntot = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = lane; i < 512; i += 64) cnt[wv][i] = 0;
        // (same wavefront: LDS operations complete in program order)
        uint32_t expect_base[4] = {0, 0, 0, 0};
        for (int sub = 0; sub < 4; ++sub) {
            const uint32_t key = rng(s) % (uint32_t)range;
            const bool valid = (rng(s) & 7u) != 0u;                  // some lanes sit a round out
            uint32_t old = 0;
            if (valid) old = atomicAdd(&cnt[wv][key >> 1], 1u << (16 * (key & 1u)));
            old = (old >> (16 * (key & 1u))) & 0xffffu;
            // reference: occurrences of `key` in earlier sub-rounds + lower lanes of this one
            uint32_t before = 0;
            for (int l = 0; l < 64; ++l) {
                const uint32_t kl = (uint32_t)__shfl((int)key, l, 64);
                const int vl = __shfl((int)valid, l, 64);
                if (vl && kl == key && l < lane) ++before;
            }
            // earlier sub-rounds: recount by replaying the generator is awkward; keep a running per-lane table instead
            (void)expect_base;
            // running count of this key before this sub-round = old - before must be the same for every lane of the key
            const uint32_t base = old - before;
            uint32_t base0 = base;
            for (int l = 0; l < 64; ++l) {
                const uint32_t kl = (uint32_t)__shfl((int)key, l, 64);
                const int vl = __shfl((int)valid, l, 64);
                const uint32_t bl = (uint32_t)__shfl((int)base, l, 64);
                if (vl && kl == key) { base0 = bl; break; }
            }
            if (valid) { ++ntot; if (base != base0 || old < before) ++nbad; }
        }
    }
    atomicAdd(bad, nbad); atomicAdd(total, ntot);
}
int main() {
    unsigned long long *d; hipMalloc(&d, 16); 
    int fails = 0;
    for (int range : {1, 2, 3, 8, 31, 64, 200, 1024}) {
        for (int waves : {1, 4, 16}) {
            hipMemset(d, 0, 16);
            if (waves == 1) k<1><<<2048, 64>>>(range, 64, d, d + 1);
            else if (waves == 4) k<4><<<2048, 256>>>(range, 64, d, d + 1);
            else k<16><<<1024, 1024>>>(range, 32, d, d + 1);
            unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("range %4d waves/wg %2d: %llu atomics, %llu out of lane order\n", range, waves, h[1], h[0]);
            if (h[0]) ++fails;
        }
    }
    printf(fails ? "LANE ORDER VIOLATED\n" : "every returning LDS atomic saw the lower lanes of its address first\n");
    return fails ? 1 : 0;
}
