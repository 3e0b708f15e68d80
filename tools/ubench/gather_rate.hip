// Micro-benchmark: how fast can ONE wave / one CU / the chip issue fragment-shaped row gathers (16 rows x 64 B per
// wave-instruction, global_load_dwordx4) when the loads are perfectly pipelined (D blocks in flight)?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/gather_rate.hip -o gpurun_out/gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KC, int D>
__global__ __launch_bounds__(64) void k_gather(const float *__restrict__ x, const int *__restrict__ idx, int rows, int C, int blocks_per_wave,
                                               float *__restrict__ out)
{
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const long w = blockIdx.x;
    const int *my = idx + w * blocks_per_wave * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 ring[D][KC];
    int iv[D + 1];
#pragma unroll
    for (int d = 0; d <= D; ++d) iv[d] = my[(d < blocks_per_wave ? d : 0) * 16 + r];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int j = 0; j < KC; ++j) ring[d][j] = *(const f32x4 *)(x + (long)iv[d] * C + 16 * j + 4 * q);
    for (int b = 0; b < blocks_per_wave; b += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // consume slot d (block b + d), refill it with block b + d + D
#pragma unroll
            for (int j = 0; j < KC; ++j) acc += ring[d][j];
            const int nb = b + d + D;
            const int nidx = iv[D];   // index for block nb was requested one step ago
            iv[D] = my[((nb + 1) < blocks_per_wave ? (nb + 1) : 0) * 16 + r];
#pragma unroll
            for (int j = 0; j < KC; ++j) ring[d][j] = *(const f32x4 *)(x + (long)nidx * C + 16 * j + 4 * q);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[w] = acc[0];
}

template <int KC, int D>
static void run(const float *x, const int *idx, int rows, int C, int waves, int bpw, float *out, const char *tag)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gather<KC, D>), dim3(waves), dim3(64), 0, 0, x, idx, rows, C, bpw, out);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gather<KC, D>), dim3(waves), dim3(64), 0, 0, x, idx, rows, C, bpw, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double loads = (double)waves * bpw * KC;
    printf("%-18s KC=%d D=%d waves=%5d blocks/wave=%4d: %7.1f us  %6.2f wave-loads/ns chip  %5.1f loads/us/wave  %6.1f GB/s\n", tag, KC, D, waves, bpw,
           ms * 1e3, loads / (ms * 1e6), (double)bpw * KC / (ms * 1e3), loads * 1024 / (ms * 1e6));
}

int main()
{
    const int rows = 50000, C = 64;
    std::vector<float> hx((size_t)rows * C, 1.f);
    const int max_blocks = 1 << 20;
    std::vector<int> hidx((size_t)max_blocks * 16);
    srand(1);
    // spatially local gather like the real tables: block b draws rows near a random centre
    for (int b = 0; b < max_blocks; ++b) {
        const int c = rand() % rows;
        for (int i = 0; i < 16; ++i) hidx[(size_t)b * 16 + i] = (c + (rand() % 64)) % rows;
    }
    float *x, *out; int *idx;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&idx, hidx.size() * 4); hipMalloc(&out, 1 << 20);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
    for (int waves : {256, 1024, 4096, 16384}) {
        const int bpw = (1 << 18) / waves > 1024 ? 1024 : (1 << 18) / waves;
        run<1, 1>(x, idx, rows, C, waves, bpw, out, "frag float4");
        run<1, 4>(x, idx, rows, C, waves, bpw, out, "frag float4");
        run<4, 1>(x, idx, rows, C, waves, bpw, out, "frag float4");
        run<4, 2>(x, idx, rows, C, waves, bpw, out, "frag float4");
        run<4, 4>(x, idx, rows, C, waves, bpw, out, "frag float4");
    }
    return 0;
}
