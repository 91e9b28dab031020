// vg_core.hip — handle lifecycle, stream and HIP-event stopwatch of the C-ABI (include/vinsgpu.h).
#include <hip/hip_runtime.h>
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

extern "C" void fe_state_destroy(FeState* s);
extern "C" void ba_graph_release(vg_handle* h);
extern "C" void ba_seq_release(vg_handle* h);
// (weak: builds without csrc/vg_rccl.hip — the CPU emulation of tests/simt — have no communicator to release)
extern "C" int vg_ba_rccl_finalize(vg_handle* h) __attribute__((weak));

extern "C" int vg_abi_version(void) { return VG_ABI_VERSION; }

extern "C" int vg_create(vg_handle** out) {
    if (!out) return VG_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return VG_ERR_NO_DEVICE;
    vg_handle* h = new vg_handle();
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return VG_ERR_HIP; }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipEventCreate(&h->ev2) != hipSuccess || hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete h;
        return VG_ERR_HIP;
    }
    *out = h;
    return VG_OK;
}

extern "C" int vg_destroy(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (vg_ba_rccl_finalize) (void)vg_ba_rccl_finalize(h);
    ba_graph_release(h);
    ba_seq_release(h);
    BaPtrs& P = h->ba.P;
    (void)hipFree(P.iarr); (void)hipFree(P.din); (void)hipFree(P.scr); (void)hipFree(P.out); (void)hipFree(P.iout);
    (void)hipFree(h->ba.dL);
    if (h->ba.h_ia) (void)hipHostFree(h->ba.h_ia);
    if (h->ba.h_di) (void)hipHostFree(h->ba.h_di);
    if (h->ba.h_pri) (void)hipHostFree(h->ba.h_pri);
    (void)hipFree(P.pri);
    h->ba.h_out.release(); h->ba.h_iout.release(); h->ba.h_mout.release(); h->ba.h_miout.release();
    (void)hipFree(P.mout); (void)hipFree(P.miout); (void)hipFree(P.mscr); (void)hipFree(P.rb1); (void)hipFree(P.rb2);
    if (h->fe) fe_state_destroy(h->fe);
    (void)hipFree(h->ransac_buf);
    (void)hipFree(h->imu_buf);
    (void)hipEventDestroy(h->ev0); (void)hipEventDestroy(h->ev1); (void)hipEventDestroy(h->ev2);
    (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join);
    (void)hipStreamDestroy(h->aux);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return VG_OK;
}

extern "C" int vg_sync(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}

extern "C" const char* vg_last_error(vg_handle* h) { return h ? h->err.c_str() : "null handle"; }
extern "C" int vg_host_register(vg_handle* h, void* p, size_t bytes) {
    if (!h || !p || !bytes) return VG_ERR_BAD_ARG;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { h->err = std::string("vg_host_register: ") + hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}
extern "C" int vg_host_unregister(vg_handle* h, void* p) {
    if (!h || !p) return VG_ERR_BAD_ARG;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { h->err = std::string("vg_host_unregister: ") + hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}
extern "C" void* vg_stream(vg_handle* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int vg_timer_start(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    hipError_t e = hipEventRecord(h->ev0, h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}

extern "C" int vg_timer_stop(vg_handle* h, float* ms) {
    if (!h || !ms) return VG_ERR_BAD_ARG;
    hipError_t e = hipEventRecord(h->ev1, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize(h->ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, h->ev0, h->ev1);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}
