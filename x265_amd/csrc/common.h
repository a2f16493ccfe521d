// common.h -- shared host-side helpers of the C-ABI library (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "../../include/x265_amd.h"

namespace x265amd {

inline int hip_status(hipError_t e) { return e == hipSuccess ? X265AMD_OK : X265AMD_ERR_HIP - (int)e; }

#define X265AMD_HIP(expr)                                   \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return ::x265amd::hip_status(e_); \
    } while (0)

// ---- per-device state ----------------------------------------------------------------------------
// The reference keeps all encoder state in its handle (X265_t, inc/x265.h:186-212).  Here the handle is
// the HIP device: everything the library remembers is kept PER DEVICE, indexed by the device that is
// current on the calling thread, and filled in by x265amd_init(device).  One host thread (or process)
// per GPU, or several threads on one GPU, can therefore use the library side by side; the mutable parts
// (the occupancy cache, the drop-in staging block) are guarded by state_mutex().
constexpr int kMaxDevices = 64;
constexpr int kOccSlots = 16;

struct OccEntry {  // occupancy of one kernel variant at one dynamic-LDS size
    const void *fn;
    size_t lds;
    int per_cu;
};

// Instrumentation switches (include/x265_amd_debug.h), per device like everything else: a setter changes the device
// that is current on the calling thread, and only calls on that device see it.
struct DebugState {
    uint32_t *me_stats = nullptr;         // x265amd_debug_set_me_stats
    bool me_timing = false;               // x265amd_debug_me_timing
    hipEvent_t me_ev[5] = {};
    int sea_lds_pad = 0, sea_waves = 0;   // x265amd_debug_sea_shape
    int tq_path = 0;                      // x265amd_debug_tq_path
    unsigned long long *tq_trace = nullptr;  // x265amd_debug_tq_trace
    int tq_trace_waves = 0;
    int tq_units_per_wg = 0;              // x265amd_debug_tq_units_per_wg
    int intra_fused = 1;                  // x265amd_debug_intra_fused
    uint32_t *intra_stats = nullptr;      // x265amd_debug_set_intra_stats
    int intra_threads = 0;                // x265amd_debug_intra_threads
};

struct DeviceState {
    std::atomic<bool> init{false};  // stored last by x265amd_init (release), read by device_state() (acquire)
    int lds_limit = 64 * 1024;  // hipDeviceAttributeMaxSharedMemoryPerBlock (160 KiB on gfx950)
    int cu_count = 256;
    void *sea_scratch = nullptr;     // device address of me_sea.hip's static scratch array ON THIS DEVICE
    size_t sea_scratch_bytes = 0;
    void *dropin_stage = nullptr;    // dropin.hip's staging block on this device
    uint32_t *host_word = nullptr;   // one page-locked, device-visible word (x265amd_init): verdicts of checking kernels
    uint32_t *host_word_dev = nullptr;  // .. as the device addresses it
    OccEntry occ[kOccSlots] = {};
    int nocc = 0;
    DebugState dbg;
};

// the current device's instrumentation switches, or a block of defaults before x265amd_init
DebugState &debug_state();

// state of the device that is current on this thread; nullptr until x265amd_init() has run for it
DeviceState *device_state();
std::mutex &state_mutex();

inline bool initialised() { return device_state() != nullptr; }
int lds_limit_bytes();
int cu_count();

// per-source-file set-up that x265amd_init(device) runs once per device, with that device current:
// the opt-in of every kernel to the device's full LDS (so no launch path touches function attributes)
// and, in me_sea.hip, the address of the static scratch array on that device
int me_kernels_device_setup(DeviceState &ds);
int me_tree_device_setup(DeviceState &ds);
int me_pu_device_setup(DeviceState &ds);
int me_sea_device_setup(DeviceState &ds);

// dynamic-LDS opt-in of one kernel: the device limit minus the kernel's own static LDS
int allow_full_lds(const void *fn, const DeviceState &ds);

inline int ilog2(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

}  // namespace x265amd
