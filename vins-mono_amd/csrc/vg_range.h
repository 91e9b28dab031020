// vg_range.h — named ranges around the C-ABI calls for rocprofv3 --marker-trace (the reference times
// its phases with TicToc + ROS_DEBUG, utility/tic_toc.h; here the same role is played by roctx ranges on the host side,
// the per-launch HIP-event profile of vg_ba_batch_run_profiled and the -DBA_PROFILE phase counters on the device side).
// The roctx library is looked up at run time: profiling support is optional, the product must load without it.
#pragma once
#include <dlfcn.h>
#include <initializer_list>
#include <mutex>

struct VgRange {
    typedef int (*PushFn)(const char*);
    typedef int (*PopFn)();
    static void resolve(PushFn& push, PopFn& pop) {
        static PushFn s_push = nullptr;
        static PopFn s_pop = nullptr;
        static std::once_flag once;               // handles may be driven from several host threads
        std::call_once(once, [] {
            for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
                void* lib = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
                if (!lib) continue;
                s_push = (PushFn)dlsym(lib, "roctxRangePushA");
                s_pop = (PopFn)dlsym(lib, "roctxRangePop");
                if (s_push && s_pop) break;
                s_push = nullptr; s_pop = nullptr;
            }
        });
        push = s_push; pop = s_pop;
    }
    PopFn pop_ = nullptr;
    explicit VgRange(const char* name) {
        PushFn push;
        resolve(push, pop_);
        if (push) push(name);
    }
    ~VgRange() { if (pop_) pop_(); }
    VgRange(const VgRange&) = delete;
    VgRange& operator=(const VgRange&) = delete;
};
#define VG_RANGE(name) VgRange _vg_range(name)
