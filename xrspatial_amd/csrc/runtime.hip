// Runtime entry points: devices, memory, streams, events.  (include/xrs_hip.h "runtime")
#include "xrs_common.h"

using namespace xrs;

extern "C" {

int xrs_version(void) { return 1; }

int xrs_last_error(char *buf, size_t buflen) {
    if (!buf || buflen == 0) return 1;
    strncpy(buf, err_buf(), buflen - 1);
    buf[buflen - 1] = 0;
    return 0;
}

int xrs_device_count(int *count) {
    if (!count) return fail("xrs_device_count: null argument");
    XRS_HIP(hipGetDeviceCount(count));
    return 0;
}

int xrs_set_device(int device) { XRS_HIP(hipSetDevice(device)); return 0; }
int xrs_get_device(int *device) { XRS_HIP(hipGetDevice(device)); return 0; }

int xrs_device_name(int device, char *buf, size_t buflen) {
    hipDeviceProp_t prop;
    XRS_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int xrs_mem_info(size_t *free_bytes, size_t *total_bytes) {
    XRS_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return 0;
}

int xrs_malloc(void **ptr_dev, size_t bytes) {
    if (!ptr_dev) return fail("xrs_malloc: null argument");
    *ptr_dev = nullptr;
    if (bytes == 0) return 0;
    XRS_HIP(hipMalloc(ptr_dev, bytes));
    return 0;
}

int xrs_free(void *ptr_dev) {
    if (ptr_dev) XRS_HIP(hipFree(ptr_dev));
    return 0;
}

int xrs_memcpy_h2d(void *dst_dev, const void *src, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemcpyAsync(dst_dev, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return 0;
}
int xrs_memcpy_d2h(void *dst, const void *src_dev, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return 0;
}
int xrs_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}
int xrs_memset(void *dst_dev, int byte_value, size_t bytes, void *stream) {
    if (bytes) XRS_HIP(hipMemsetAsync(dst_dev, byte_value, bytes, as_stream(stream)));
    return 0;
}

int xrs_stream_create(void **stream) {
    hipStream_t s;
    XRS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return 0;
}
int xrs_stream_destroy(void *stream) { if (stream) XRS_HIP(hipStreamDestroy(as_stream(stream))); return 0; }
int xrs_stream_sync(void *stream) { XRS_HIP(hipStreamSynchronize(as_stream(stream))); return 0; }
int xrs_device_sync(void) { XRS_HIP(hipDeviceSynchronize()); return 0; }

int xrs_event_create(void **event) {
    hipEvent_t e;
    XRS_HIP(hipEventCreate(&e));
    *event = e;
    return 0;
}
int xrs_event_destroy(void *event) { if (event) XRS_HIP(hipEventDestroy((hipEvent_t)event)); return 0; }
int xrs_event_record(void *event, void *stream) { XRS_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream))); return 0; }
int xrs_event_sync(void *event) { XRS_HIP(hipEventSynchronize((hipEvent_t)event)); return 0; }
int xrs_event_elapsed_ms(void *start_event, void *stop_event, float *ms) {
    XRS_HIP(hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event));
    return 0;
}

}  // extern "C"
