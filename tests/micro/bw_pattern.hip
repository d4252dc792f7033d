// Micro-benchmark: HBM bandwidth of the GEMM's activation access patterns (not part of the product).
//   read  A: per workgroup 256 rows x 64 B at a 1 KB row stride, 16 consecutive "K stages" (what gemm_f16x3 does)
//   read  B: the same bytes laid out so that a stage is one contiguous 16 KB block
//   write A: every lane writes 64 contiguous bytes at a 1 KB stride (row-major SH8 epilogue)
//   write B: lanes write consecutive 64-byte units (tiled layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_strided(const u32x4* in, u32x4* out, int row_u4, int stages) {
    const int t = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i, r = idx >> 2, q = idx & 3;
            acc ^= in[(row0 + r) * row_u4 + s * 4 + q];
        }
    }
    if (acc[0] == 0x12345678u) out[blockIdx.x * 256 + t] = acc;
}
__global__ __launch_bounds__(256) void read_tiled(const u32x4* in, u32x4* out, int row_u4, int stages) {
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * 256 * row_u4;
    u32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc ^= in[base + (long long)s * 1024 + t + 256 * i];
    }
    if (acc[0] == 0x12345678u) out[blockIdx.x * 256 + t] = acc;
}
// 4 waves x 64 lanes: lane = row, each lane writes 4 x 16 B of chunk c
__global__ __launch_bounds__(256) void write_strided(u32x4* out, int row_u4, int chunks) {
    const int t = threadIdx.x;
    const long long row = (long long)blockIdx.x * 256 + t;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[row * row_u4 + c * 4 + q] = u32x4{(unsigned)t, (unsigned)c, (unsigned)q, 1u};
    }
}
__global__ __launch_bounds__(256) void write_tiled(u32x4* out, int row_u4, int chunks) {
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * 256 * row_u4;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[base + (long long)c * 1024 + t * 4 + q] = u32x4{(unsigned)t, (unsigned)c, (unsigned)q, 1u};
    }
}

// every store instruction of a wave covers 1 KB contiguous
__global__ __launch_bounds__(256) void write_coalesced(u32x4* out, int row_u4, int chunks) {
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * 256 * row_u4;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[base + (long long)c * 1024 + q * 256 + t] = u32x4{(unsigned)t, (unsigned)c, (unsigned)q, 1u};
    }
}
// same, 4-byte stores (what a plain fp32 epilogue does)
__global__ __launch_bounds__(256) void write_coalesced_b32(unsigned* out, int row_u4, int chunks) {
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * 256 * row_u4 * 4;
    for (int c = 0; c < chunks * 16; ++c) out[base + (long long)c * 256 + t] = (unsigned)(t + c);
}
__global__ __launch_bounds__(256) void copy_coalesced(const u32x4* in, u32x4* out, int row_u4, int chunks) {
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * 256 * row_u4;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) out[base + (long long)c * 1024 + q * 256 + t] = in[base + (long long)c * 1024 + q * 256 + t];
    }
}

// proposed epilogue: after a quad transpose, 4 lanes write one row's 64 B and lanes L / L+32 the two halves of a
// 128-byte line: every store instruction writes 8 full lines (row stride 1 KB)
__global__ __launch_bounds__(256) void write_quad(u32x4* out, int row_u4, int chunks) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int h = lane >> 5, j = (lane & 31) >> 2, q = lane & 3;
    const long long row0 = (long long)blockIdx.x * 256 + wave * 64;
    for (int c = 0; c < chunks / 2; ++c)       // 128-byte column chunks
        for (int half = 0; half < 2; ++half)   // two 32-row MFMA tiles per wave
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const long long row = row0 + half * 32 + 4 * j + s;
                out[row * row_u4 + c * 8 + h * 4 + q] = u32x4{(unsigned)t, (unsigned)c, (unsigned)s, 1u};
            }
}

int main() {
    const int rows = 65536 * 2, row_u4 = 64;  // 128 K rows x 1 KB = 134 MB
    const size_t bytes = (size_t)rows * row_u4 * 16;
    u32x4 *a, *b, *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 1 << 24);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto f) {
        for (int i = 0; i < 3; ++i) f();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-16s %8.1f us  %6.2f TB/s\n", name, ms * 50, bytes / (ms / 20 * 1e-3) / 1e12);
    };
    const int grid = rows / 256;
    // alternate the buffers so that the 256 MB infinity cache cannot serve a pass from the previous one
    int flip = 0;
    time("read strided", [&] { hipLaunchKernelGGL(read_strided, dim3(grid), dim3(256), 0, 0, (flip ^= 1) ? a : b, o, row_u4, 16); });
    time("read tiled", [&] { hipLaunchKernelGGL(read_tiled, dim3(grid), dim3(256), 0, 0, (flip ^= 1) ? a : b, o, row_u4, 16); });
    time("write strided", [&] { hipLaunchKernelGGL(write_strided, dim3(grid), dim3(256), 0, 0, (flip ^= 1) ? a : b, row_u4, 16); });
    time("write tiled", [&] { hipLaunchKernelGGL(write_tiled, dim3(grid), dim3(256), 0, 0, (flip ^= 1) ? a : b, row_u4, 16); });
    time("write quad", [&] { hipLaunchKernelGGL(write_quad, dim3(grid), dim3(256), 0, 0, (flip ^= 1) ? a : b, row_u4, 16); });
    time("write coalesced", [&] { hipLaunchKernelGGL(write_coalesced, dim3(grid), dim3(256), 0, 0, (flip ^= 1) ? a : b, row_u4, 16); });
    time("write coal. b32", [&] { hipLaunchKernelGGL(write_coalesced_b32, dim3(grid), dim3(256), 0, 0, (unsigned*)((flip ^= 1) ? a : b), row_u4, 16); });
    time("hipMemsetAsync", [&] { hipMemsetAsync((flip ^= 1) ? a : b, 3, bytes, 0); });
    time("copy (r+w bytes/2)", [&] { hipLaunchKernelGGL(copy_coalesced, dim3(grid), dim3(256), 0, 0, a, b, row_u4, 16); });
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
