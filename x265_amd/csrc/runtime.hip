// runtime.hip -- device selection, error strings, memory helpers of the C-ABI
#include <string.h>
#include <algorithm>

#include "common.h"

namespace x265amd {

static std::mutex g_mu;
static DeviceState g_dev[kMaxDevices];

std::mutex &state_mutex() { return g_mu; }

DeviceState *device_state()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    DeviceState *ds = &g_dev[dev];
    return ds->init.load(std::memory_order_acquire) ? ds : nullptr;  // stored last by x265amd_init, never cleared
}
DebugState &debug_state()
{
    static DebugState defaults;  // never written: the setters return early without a device
    DeviceState *ds = device_state();
    return ds ? ds->dbg : defaults;
}
int lds_limit_bytes()
{
    const DeviceState *ds = device_state();
    return ds ? ds->lds_limit : 64 * 1024;
}
int cu_count()
{
    const DeviceState *ds = device_state();
    return ds ? ds->cu_count : 256;
}

int allow_full_lds(const void *fn, const DeviceState &ds)
{
    hipFuncAttributes fa;
    X265AMD_HIP(hipFuncGetAttributes(&fa, fn));
    const int dyn = ds.lds_limit - (int)fa.sharedSizeBytes;
    if (dyn <= 0) return X265AMD_ERR_LDS;
    X265AMD_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    return X265AMD_OK;
}

}  // namespace x265amd

using namespace x265amd;

extern "C" int x265amd_version(void) { return X265AMD_VERSION; }

extern "C" int x265amd_init(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return X265AMD_ERR_NO_DEVICE;
    if (device < 0 || device >= n || device >= kMaxDevices) return X265AMD_ERR_NO_DEVICE;
    X265AMD_HIP(hipSetDevice(device));
    DeviceState &ds = g_dev[device];
    if (ds.init.load(std::memory_order_acquire)) return X265AMD_OK;  // idempotent; the device is current on this thread now
    hipDeviceProp_t prop;
    X265AMD_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return X265AMD_ERR_ARCH;  // kernels are built for gfx950 only
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
        ds.lds_limit = lds;
    if (prop.multiProcessorCount > 0) ds.cu_count = prop.multiProcessorCount;
    // once per device, with that device current: LDS opt-ins and the scratch array's address
    int st = me_kernels_device_setup(ds);
    if (st == X265AMD_OK) st = me_tree_device_setup(ds);
    if (st == X265AMD_OK) st = me_pu_device_setup(ds);
    if (st == X265AMD_OK) st = me_sea_device_setup(ds);
    if (st != X265AMD_OK) return st;
    if (!ds.host_word) {
        void *hp = nullptr, *dp = nullptr;
        X265AMD_HIP(hipHostMalloc(&hp, 64, hipHostMallocMapped));
        X265AMD_HIP(hipHostGetDevicePointer(&dp, hp, 0));
        ds.host_word = (uint32_t *)hp;
        ds.host_word_dev = (uint32_t *)dp;
    }
    ds.init.store(true, std::memory_order_release);  // everything above is visible to a thread that sees it
    return X265AMD_OK;
}

extern "C" const char *x265amd_strerror(int status)
{
    switch (status) {
    case X265AMD_OK: return "ok";
    case X265AMD_ERR_ARG: return "invalid argument (geometry, size, alignment or null pointer)";
    case X265AMD_ERR_NO_DEVICE: return "no such HIP device";
    case X265AMD_ERR_ARCH: return "device is not gfx950 (MI355X); this library has no other code path";
    case X265AMD_ERR_NOT_INIT: return "x265amd_init() has not been called for the current device";
    case X265AMD_ERR_LDS: return "search window does not fit in LDS";
    case X265AMD_ERR_WORKSPACE: return "job exceeds the static scratch of x265amd_me_fullsearch_sea_static; use x265amd_me_fullsearch_sea with a workspace";
    case X265AMD_ERR_BUFFER: return "output buffer too small";
    default: break;
    }
    if (status <= X265AMD_ERR_HIP) return hipGetErrorString((hipError_t)(X265AMD_ERR_HIP - status));
    return "unknown status";
}

extern "C" int x265amd_malloc(void **d_ptr, size_t bytes)
{
    if (!d_ptr) return X265AMD_ERR_ARG;
    X265AMD_HIP(hipMalloc(d_ptr, bytes));
    return X265AMD_OK;
}
extern "C" int x265amd_free(void *d_ptr)
{
    X265AMD_HIP(hipFree(d_ptr));
    return X265AMD_OK;
}
extern "C" int x265amd_malloc_host(void **h_ptr, size_t bytes)
{
    if (!h_ptr) return X265AMD_ERR_ARG;
    X265AMD_HIP(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return X265AMD_OK;
}
extern "C" int x265amd_free_host(void *h_ptr)
{
    X265AMD_HIP(hipHostFree(h_ptr));
    return X265AMD_OK;
}
extern "C" int x265amd_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    X265AMD_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return X265AMD_OK;
}
extern "C" int x265amd_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    X265AMD_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return X265AMD_OK;
}
// x265amd_copy_to_host: the copy as a kernel of the stream.  T = uint4 when everything is 16-byte aligned.
template <class T>
__global__ __launch_bounds__(256) void copy_to_host_kernel(T *__restrict__ dst, const T *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
extern "C" int x265amd_copy_to_host(void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!initialised()) return X265AMD_ERR_NOT_INIT;
    if (!h_dst || !d_src || (bytes & 3) || ((uintptr_t)h_dst & 3) || ((uintptr_t)d_src & 3)) return X265AMD_ERR_ARG;
    if (bytes == 0) return X265AMD_OK;
    void *mapped = nullptr;  // fails for memory that is not page-locked and mapped (x265amd_malloc_host's is)
    if (hipHostGetDevicePointer(&mapped, h_dst, 0) != hipSuccess || !mapped) {
        (void)hipGetLastError();
        return X265AMD_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (((bytes | (uintptr_t)mapped | (uintptr_t)d_src) & 15) == 0) {
        const size_t n = bytes / 16;
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
        hipLaunchKernelGGL(copy_to_host_kernel<uint4>, dim3(grid), dim3(256), 0, st, (uint4 *)mapped, (const uint4 *)d_src, n);
    } else {
        const size_t n = bytes / 4;
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
        hipLaunchKernelGGL(copy_to_host_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, (uint32_t *)mapped, (const uint32_t *)d_src, n);
    }
    return hip_status(hipGetLastError());
}
extern "C" int x265amd_stream_sync(void *stream)
{
    X265AMD_HIP(hipStreamSynchronize((hipStream_t)stream));
    return X265AMD_OK;
}

extern "C" int x265amd_stream_create(void **stream)
{
    if (!stream) return X265AMD_ERR_ARG;
    hipStream_t s;
    X265AMD_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return X265AMD_OK;
}
extern "C" int x265amd_stream_destroy(void *stream)
{
    if (!stream) return X265AMD_ERR_ARG;
    X265AMD_HIP(hipStreamDestroy((hipStream_t)stream));
    return X265AMD_OK;
}

extern "C" int x265amd_chroma_qp(int qp)
{
    // g_aucChromaScale (src/table.cpp:166-171): identity below 30, then the draft's chroma mapping
    static const unsigned char tail[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36,
                                           36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
    if (qp < 0 || qp > 51) return X265AMD_ERR_ARG;
    return qp < 30 ? qp : tail[qp - 30];
}
