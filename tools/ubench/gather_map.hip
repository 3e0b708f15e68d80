// Micro-benchmark: does the LANE -> (row, 16-byte piece) mapping of a gathered 16-row x 64-byte fragment matter?
//   MAP 0: MFMA-fragment shape, lane l reads row l & 15, piece l >> 4  (consecutive lanes = different rows)   <- the kernels
//   MAP 1: row-major, lane l reads row l >> 2, piece l & 3               (a quad of lanes = 64 contiguous bytes)
//   MAP 2: KC = 4 only: load j reads rows 4j .. 4j+3, lane l row 4j + (l >> 4), piece l & 15 (16 lanes = 256 contiguous bytes)
// Same rows, same bytes, same cache lines per block in every mapping; D blocks in flight per wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/gather_map.hip -o tools/ubench/gather_map.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KC, int D, int MAP>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ x, const int *__restrict__ idx, int C, int blocks_per_wave,
                                                float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int *my = idx + w * blocks_per_wave * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 ring[D][KC];
    auto issue = [&](int d, int b) {
        const int bb = b < blocks_per_wave ? b : 0;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            int row, off;
            if (MAP == 0) { row = my[bb * 16 + (lane & 15)]; off = 16 * j + 4 * (lane >> 4); }
            else if (MAP == 1) { row = my[bb * 16 + (lane >> 2)]; off = 16 * j + 4 * (lane & 3); }
            else { row = my[bb * 16 + 4 * j + (lane >> 4)]; off = 4 * (lane & 15); }
            ring[d][j] = *(const f32x4 *)(x + (long)row * C + off);
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    for (int b = 0; b < blocks_per_wave; b += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int j = 0; j < KC; ++j) acc += ring[d][j];
            issue(d, b + d + D);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[w] = acc[0];
}

template <int KC, int D, int MAP>
static void run(const float *x, const int *idx, int C, int waves, int bpw, float *out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wpg = 4;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_gather<KC, D, MAP>), dim3(waves / wpg), dim3(64 * wpg), 0, 0, x, idx, C, bpw, out);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gather<KC, D, MAP>), dim3(waves / wpg), dim3(64 * wpg), 0, 0, x, idx, C, bpw, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double loads = (double)waves * bpw * KC;
    // cycles per wave-load and CU at 2.4 GHz with 256 CUs busy
    printf("MAP %d KC=%d D=%d waves=%5d blocks/wave=%4d: %7.1f us  %6.2f wave-loads/ns chip = %5.1f cycles per load and CU  %6.1f GB/s\n", MAP, KC, D, waves, bpw,
           ms * 1e3, loads / (ms * 1e6), 2.4 * 256.0 / (loads / (ms * 1e6)), loads * 1024 / (ms * 1e6));
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 50000, C = 64;
    std::vector<float> hx((size_t)rows * C, 1.f);
    const int max_blocks = 1 << 20;
    std::vector<int> hidx((size_t)max_blocks * 16);
    srand(1);
    for (int b = 0; b < max_blocks; ++b) {
        const int c = rand() % rows;
        for (int i = 0; i < 16; ++i) hidx[(size_t)b * 16 + i] = (c + (rand() % 64)) % rows;
    }
    float *x, *out; int *idx;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&idx, hidx.size() * 4); hipMalloc(&out, 1 << 22);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
    printf("table: %d rows x %d floats = %.1f MB\n", rows, C, rows * C * 4 / 1e6);
    for (int waves : {4096, 16384}) {
        const int bpw = (1 << 20) / waves > 256 ? 256 : (1 << 20) / waves;
        run<1, 4, 0>(x, idx, C, waves, bpw, out);
        run<1, 4, 1>(x, idx, C, waves, bpw, out);
        run<4, 2, 0>(x, idx, C, waves, bpw, out);
        run<4, 2, 1>(x, idx, C, waves, bpw, out);
        run<4, 2, 2>(x, idx, C, waves, bpw, out);
        run<2, 4, 0>(x, idx, C, waves, bpw, out);
        run<2, 4, 1>(x, idx, C, waves, bpw, out);
    }
    return 0;
}
