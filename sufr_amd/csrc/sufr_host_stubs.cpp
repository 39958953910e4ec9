// sufr_host_stubs.cpp -- the device side of the C ABI as "no device" stubs, for the HOST-ONLY sanitizer build
// (`make asan` -> _build/libsufr_host_asan.so: sufr_io.cpp + sufr_query.cpp + this file, -fsanitize=address,undefined).
//
// What the sanitizer build is for: the parallel FASTA / FASTQ parser (gzip, bzip2, xz), the .sufr writer and the parser
// of untrusted .sufr files run on the host and never touch the GPU; tests/test_sanitized_host.py runs the reader, writer
// and query tests and a fuzz loop over damaged inputs against this library under AddressSanitizer + UBSan.
// Every entry point that needs a device answers SUFR_HIP_E_NO_DEVICE here -- the product library (libsufr_hip.so) never
// contains this file, and GPU AddressSanitizer is not used anywhere.
#include <stdint.h>
#include <stddef.h>
#include <string>

#include <hip/hip_runtime_api.h>

#include "../../include/sufr_hip.h"
#include "../../include/sufr_query.h"

struct sufr_hip_ctx { std::string err; };

namespace {
const char* const NO_DEVICE = "host-only sanitizer build: no HIP device (libsufr_host_asan.so has no device code)";
int no_device(sufr_hip_ctx* ctx) { if (ctx) ctx->err = NO_DEVICE; return SUFR_HIP_E_NO_DEVICE; }
}

extern "C" {

// ---- HIP runtime entry points sufr_io.cpp references (pinned staging buffers, copy streams of the writers, the per-device arrays of the multi-context windowed create) ----------
hipError_t hipHostMalloc(void** p, size_t, unsigned int) { if (p) *p = nullptr; return hipErrorNoDevice; }
hipError_t hipHostFree(void*) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "no device"; }
hipError_t hipMalloc(void** p, size_t) { if (p) *p = nullptr; return hipErrorNoDevice; }
hipError_t hipFree(void*) { return hipSuccess; }
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind) { return hipErrorNoDevice; }
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t) { return hipErrorNoDevice; }
hipError_t hipSetDevice(int) { return hipErrorNoDevice; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { if (s) *s = nullptr; return hipErrorNoDevice; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipErrorNoDevice; }

// ---- include/sufr_hip.h: contexts and builds ------------------------------------------------------------------------
int sufr_hip_abi_version(void) { return SUFR_HIP_ABI_VERSION; }
int sufr_hip_device_count(void) { return 0; }
sufr_hip_ctx* sufr_hip_create(int) { return nullptr; }
void sufr_hip_destroy(sufr_hip_ctx* ctx) { delete ctx; }
const char* sufr_hip_last_error(const sufr_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : NO_DEVICE; }
int sufr_hip_set_stream(sufr_hip_ctx* ctx, void*) { return no_device(ctx); }
int sufr_hip_synchronize(sufr_hip_ctx* ctx) { return no_device(ctx); }
int sufr_hip_set_window(sufr_hip_ctx* ctx, uint64_t, uint64_t) { return no_device(ctx); }
int sufr_hip_set_window_retry(sufr_hip_ctx* ctx, uint64_t) { return no_device(ctx); }
int sufr_hip_set_array_budget(sufr_hip_ctx* ctx, uint64_t) { return no_device(ctx); }
uint64_t sufr_hip_window_repairs(const sufr_hip_ctx*) { return 0; }
int sufr_hip_sort_device_u32(sufr_hip_ctx* ctx, const void*, uint64_t, uint32_t, uint64_t, const char*, uint64_t, uint64_t,
                             uint32_t, uint32_t, void*, void*, uint64_t, uint64_t*, sufr_hip_stats*) { return no_device(ctx); }
int sufr_hip_sort_device_u64(sufr_hip_ctx* ctx, const void*, uint64_t, uint32_t, uint64_t, const char*, uint64_t, uint64_t,
                             uint32_t, uint32_t, void*, void*, uint64_t, uint64_t*, sufr_hip_stats*) { return no_device(ctx); }
int sufr_hip_stitch_device_u32(sufr_hip_ctx* ctx, uint64_t, const uint64_t*, uint32_t, uint32_t, void*) { return no_device(ctx); }
int sufr_hip_stitch_device_u64(sufr_hip_ctx* ctx, uint64_t, const uint64_t*, uint32_t, uint32_t, void*) { return no_device(ctx); }
int sufr_hip_build_u32(sufr_hip_ctx* ctx, const uint8_t*, uint64_t, uint32_t, uint64_t, const char*, uint64_t, uint64_t, uint8_t*,
                       uint32_t*, uint32_t*, uint64_t, uint64_t*, sufr_hip_stats*) { return no_device(ctx); }
int sufr_hip_build_u64(sufr_hip_ctx* ctx, const uint8_t*, uint64_t, uint32_t, uint64_t, const char*, uint64_t, uint64_t, uint8_t*,
                       uint64_t*, uint64_t*, uint64_t, uint64_t*, sufr_hip_stats*) { return no_device(ctx); }

// ---- internal hooks of sufr_io.cpp into the pipeline (sufr_capi.inc) ------------------------------------------------
void sufr_hip_set_error_(sufr_hip_ctx* ctx, const char* msg) { if (ctx) ctx->err = msg ? msg : ""; }
int sufr_hip_is_wide_(const sufr_hip_ctx*, uint64_t) { return 0; }
int sufr_hip_ctx_device_(const sufr_hip_ctx*) { return -1; }
uint64_t sufr_hip_array_budget_(const sufr_hip_ctx*) { return 0; }
void sufr_hip_release_build_arrays_(sufr_hip_ctx*, int) {}
void sufr_hip_release_wide_arrays_(sufr_hip_ctx*) {}
int sufr_hip_build_resident_(sufr_hip_ctx* ctx, const uint8_t*, uint64_t, uint32_t, uint64_t, const char*, uint32_t, uint32_t,
                             uint64_t*, sufr_hip_stats*, int*, const void**, const void**, const void**) { return no_device(ctx); }
int sufr_hip_resident_ends_(sufr_hip_ctx* ctx, uint64_t, uint64_t*, uint64_t*) { return no_device(ctx); }
int sufr_hip_resident_stitch_(sufr_hip_ctx* ctx, uint64_t, uint64_t, uint64_t*) { return no_device(ctx); }
int sufr_hip_resident_arrays_(sufr_hip_ctx* ctx, int*, const void**, const void**, const void**) { return no_device(ctx); }

// ---- include/sufr_query.h: the device search -----------------------------------------------------------------------
struct sufr_hip_index { int unused; };
int sufr_hip_index_load(sufr_hip_ctx* ctx, const sufr_file*, sufr_hip_index** out) { if (out) *out = nullptr; return no_device(ctx); }
int sufr_hip_index_wrap(sufr_hip_ctx* ctx, const void*, uint64_t, const void*, uint64_t, uint32_t, uint64_t, const char*,
                        sufr_hip_index** out) { if (out) *out = nullptr; return no_device(ctx); }
void sufr_hip_index_free(sufr_hip_index*) {}
int sufr_hip_index_width(const sufr_hip_index*) { return 0; }
int sufr_hip_search_batch(sufr_hip_ctx* ctx, const sufr_hip_index*, const uint8_t*, const uint64_t*, uint64_t, int, uint64_t,
                          uint64_t*, uint64_t*) { return no_device(ctx); }
int sufr_hip_search_batch_device(sufr_hip_ctx* ctx, const sufr_hip_index*, const void*, const void*, uint64_t, int, uint64_t,
                                 void*, void*) { return no_device(ctx); }
int sufr_hip_locate_batch_device(sufr_hip_ctx* ctx, const sufr_hip_index*, const void*, const void*, uint64_t, uint64_t, void*,
                                 void*, uint64_t, uint64_t*) { return no_device(ctx); }

}  // extern "C"
