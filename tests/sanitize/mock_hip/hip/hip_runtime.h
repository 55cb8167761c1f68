// TEST INFRASTRUCTURE: a host-memory stand-in for the 13 HIP runtime calls picha_amd/csrc/dispatch.cpp makes, so that the
// dispatcher's queue / slot / thread logic can run under ThreadSanitizer and AddressSanitizer on a box without a GPU
// (GPU sanitizers are not available on this pool).  "Device" memory is malloc'd, copies are memcpy, streams and events
// complete immediately.  Never part of the product build.
#pragma once
#include <cstdlib>
#include <cstring>
typedef int hipError_t;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
inline const char *hipGetErrorString(hipError_t) { return "mock hip error"; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (void *)1; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (void *)1; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
