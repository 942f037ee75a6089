// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this library uses (MI355X_MICROARCH.md,
// section HBM: the counter reads half of the bytes of a 16 B/lane streaming read; other widths are uncalibrated).
// Reads a 1 GiB buffer once with 4 B/lane, 16 B/lane and 1 B/lane loads; run under `rocprofv3 --pmc FETCH_SIZE`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void read4(const uint32_t* __restrict__ p, size_t n, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read16(const uint4* __restrict__ p, size_t n, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read1(const uint8_t* __restrict__ p, size_t n, uint32_t* out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t bytes = 1ull << 30;
    void* d; uint32_t* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4);
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    read4<<<4096, 256>>>((const uint32_t*)d, bytes / 4, o);
    read16<<<4096, 256>>>((const uint4*)d, bytes / 16, o);
    read1<<<4096, 256>>>((const uint8_t*)d, bytes / 4, o); // first 256 MiB only
    hipDeviceSynchronize();
    printf("bytes read: read4 %zu read16 %zu read1 %zu\n", bytes, bytes, bytes / 4);
    return 0;
}
