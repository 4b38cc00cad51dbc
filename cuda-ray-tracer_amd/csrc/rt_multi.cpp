// rt_multi.cpp -- several GPUs of one node behind ONE call (libmi355rt_multi.so; declared in include/mi355rt.h).
//
// The reference is single-GPU (src/update-cuda.cu:160-190); BASELINE.json's north star adds row tiling across the GPUs
// of a node with a gather to one GPU as the only exchange step.  This layer puts that underneath the
// update() boundary (include/update.h:6-8): one process, one context per (device, part), rows band-cyclic over all
// contexts, every device renders its parts on its own stream, finished parts travel to the root device over RCCL
// (ncclSend / ncclRecv in one group per part, xGMI point to point) on a second stream per device while the next part
// renders, and the root restores row order with rt_assemble.  No collective touches the rendering itself.
//
//   transport   distinct devices            RCCL: ncclCommInitAll over the device list, one communicator per device
//               the same device n times     device-to-device copies on the comm streams: the whole choreography (bands,
//                                           offsets, events, reassembly) on a one-GPU box, without RCCL
//               one device, SELF_EXCHANGE   RCCL with one rank that sends its rows to itself: the RCCL calls on a one-GPU box
//
// Everything is enqueue-only unless timing is requested; rt_multi_wait() / rt_multi_stream() order later work.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mi355rt.h"

namespace {

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    rt_set_last_error(buf);
    return code;
}

#define M_HIP(call)                                                                                      \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define M_NCCL(call)                                                                                      \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) return fail(RT_ERR_DEVICE, "%s failed: %s", #call, ncclGetErrorString(r_)); \
    } while (0)

enum Transport { DIRECT = 0, LOCAL_COPY = 1, RCCL = 2 };

// every entry point moves the calling thread from device to device; it leaves with the device it came with
struct DeviceRestore {
    int d = -1;
    DeviceRestore() { if (hipGetDevice(&d) != hipSuccess) d = -1; }
    ~DeviceRestore() { if (d >= 0) (void) hipSetDevice(d); }
    DeviceRestore(const DeviceRestore &) = delete;
    DeviceRestore &operator=(const DeviceRestore &) = delete;
};

} // namespace

struct rt_multi {
    uint32_t n = 0, parts = 1, world = 1; // devices, parts per device, contexts = n * parts
    uint32_t width = 0, height = 0;
    size_t pixel_bytes = 16, slot_bytes = 0, full_bytes = 0;
    Transport transport = DIRECT;
    bool self_exchange = false;
    bool bandwise = false;            // RT_MULTI_BANDWISE: rows travel band by band straight to their place in the full frame; no rank-major slots, no rt_assemble
    uint32_t band_rows = 16;
    std::vector<uint32_t> rows;       // [world] local rows of context q (bandwise)
    std::vector<int> dev;              // [n]
    std::vector<rt_ctx *> ctx;         // [world], context q = part * n + r lives on device r and is rank q of `world`
    std::vector<hipStream_t> s_render, s_comm; // [n]
    std::vector<void *> local;         // [world] rows of context q on its own device (NULL where it renders into the root's slot)
    std::vector<hipEvent_t> ev_rendered, ev_sent; // [world]
    std::vector<ncclComm_t> comm;      // [n] (RCCL only)
    void *gathered = nullptr;          // root device: [world][max_local_rows][width] pixels, rank-major
    void *full = nullptr;              // root device: [height][width] pixels
    hipEvent_t ev_gathered = nullptr, ev_assembled = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    bool have_assembled = false;
    void *last_full = nullptr;         // where the last frame went (rt_render_multi's root_full_fb, or `full`): what rt_multi_download reads
    bool in_flight_failed = false;     // a frame failed after part of it had been enqueued: events and receive slots are in an unknown state
};

extern "C" int rt_multi_destroy(rt_multi *m)
{
    if (!m) return RT_OK;
    DeviceRestore restore;
    for (uint32_t r = 0; r < m->n; r++) {
        (void) hipSetDevice(m->dev[r]);
        if (r < m->s_render.size() && m->s_render[r]) (void) hipStreamSynchronize(m->s_render[r]);
        if (r < m->s_comm.size() && m->s_comm[r]) (void) hipStreamSynchronize(m->s_comm[r]);
    }
    for (size_t r = 0; r < m->comm.size(); r++)
        if (m->comm[r]) (void) ncclCommDestroy(m->comm[r]);
    for (uint32_t q = 0; q < m->ctx.size(); q++) {
        const uint32_t r = q % m->n;
        (void) hipSetDevice(m->dev[r]);
        if (m->ctx[q]) (void) rt_destroy(m->ctx[q]);
        if (q < m->local.size() && m->local[q]) (void) hipFree(m->local[q]);
        if (q < m->ev_rendered.size() && m->ev_rendered[q]) (void) hipEventDestroy(m->ev_rendered[q]);
        if (q < m->ev_sent.size() && m->ev_sent[q]) (void) hipEventDestroy(m->ev_sent[q]);
    }
    if (m->n) (void) hipSetDevice(m->dev[0]);
    if (m->gathered) (void) hipFree(m->gathered);
    if (m->full) (void) hipFree(m->full);
    for (hipEvent_t e : {m->ev_gathered, m->ev_assembled, m->ev_t0, m->ev_t1})
        if (e) (void) hipEventDestroy(e);
    for (uint32_t r = 0; r < m->n; r++) {
        (void) hipSetDevice(m->dev[r]);
        if (r < m->s_render.size() && m->s_render[r]) (void) hipStreamDestroy(m->s_render[r]);
        if (r < m->s_comm.size() && m->s_comm[r]) (void) hipStreamDestroy(m->s_comm[r]);
    }
    delete m;
    return RT_OK;
}

static int create_impl(rt_multi *m, const rt_scene_desc *sd, const int *devices, uint32_t n, uint32_t band_rows, uint32_t parts, uint32_t flags, uint32_t format)
{
    m->n = n;
    m->parts = parts;
    m->world = n * parts;
    m->width = sd->width;
    m->height = sd->height;
    m->pixel_bytes = format == RT_FMT_RGBA8 ? 4 : 16;
    m->self_exchange = (flags & RT_MULTI_SELF_EXCHANGE) != 0;
    m->bandwise = (flags & RT_MULTI_BANDWISE) != 0;
    m->band_rows = band_rows;
    m->dev.assign(devices, devices + n);
    bool all_same = true, all_distinct = true;
    for (uint32_t a = 0; a < n; a++)
        for (uint32_t b = a + 1; b < n; b++) {
            if (devices[a] == devices[b]) all_distinct = false;
            else all_same = false;
        }
    if (n > 1 && !all_same && !all_distinct) return fail(RT_ERR_INVALID, "rt_create_multi: the device list must be all distinct (RCCL) or one device repeated (rehearsal on one GPU)");
    m->transport = n == 1 ? (m->self_exchange ? RCCL : DIRECT) : (all_distinct ? RCCL : LOCAL_COPY);

    m->ctx.assign(m->world, nullptr);
    m->local.assign(m->world, nullptr);
    m->ev_rendered.assign(m->world, nullptr);
    m->ev_sent.assign(m->world, nullptr);
    m->s_render.assign(n, nullptr);
    m->s_comm.assign(n, nullptr);
    for (uint32_t r = 0; r < n; r++) {
        M_HIP(hipSetDevice(devices[r]));
        M_HIP(hipStreamCreateWithFlags(&m->s_render[r], hipStreamNonBlocking));
        M_HIP(hipStreamCreateWithFlags(&m->s_comm[r], hipStreamNonBlocking));
    }
    uint32_t max_rows = 0;
    for (uint32_t q = 0; q < m->world; q++) {
        const uint32_t r = q % n;
        rt_config cfg{};
        cfg.device = devices[r];
        cfg.rank = q;
        cfg.world = m->world;
        cfg.band_rows = band_rows;
        cfg.flags = flags & ~(RT_MULTI_SELF_EXCHANGE | RT_MULTI_BANDWISE);
        cfg.format = format;
        int rc = rt_create(&m->ctx[q], sd, &cfg);
        if (rc != RT_OK) return rc;
        M_HIP(hipSetDevice(devices[r]));
        M_HIP(hipEventCreateWithFlags(&m->ev_rendered[q], hipEventDisableTiming));
        M_HIP(hipEventCreateWithFlags(&m->ev_sent[q], hipEventDisableTiming));
        if (q == 0) rt_max_local_rows(m->ctx[0], &max_rows);
    }
    m->rows.assign(m->world, 0);
    for (uint32_t q = 0; q < m->world; q++) rt_local_rows(m->ctx[q], &m->rows[q]);
    if (m->bandwise && m->world == 1 && m->transport == DIRECT) m->bandwise = false; // (one context renders in place: nothing travels)
    m->slot_bytes = (size_t) max_rows * sd->width * m->pixel_bytes;
    m->full_bytes = (size_t) sd->height * sd->width * m->pixel_bytes;
    M_HIP(hipSetDevice(devices[0]));
    M_HIP(hipMalloc(&m->full, m->full_bytes ? m->full_bytes : 16));
    if ((m->world > 1 || m->transport == RCCL) && !m->bandwise) M_HIP(hipMalloc(&m->gathered, m->slot_bytes * m->world + 16));
    for (uint32_t q = 0; q < m->world; q++) { // rows that have to travel get a buffer on their own device
        const uint32_t r = q % n;
        const bool travels = m->bandwise || (m->transport == RCCL ? (r != 0 || m->self_exchange) : (m->transport == LOCAL_COPY && r != 0)); // (bandwise: the root's own rows are strided in the frame too)
        if (!travels) continue;
        M_HIP(hipSetDevice(devices[r]));
        M_HIP(hipMalloc(&m->local[q], m->slot_bytes ? m->slot_bytes : 16));
    }
    M_HIP(hipSetDevice(devices[0]));
    M_HIP(hipEventCreateWithFlags(&m->ev_gathered, hipEventDisableTiming));
    M_HIP(hipEventCreateWithFlags(&m->ev_assembled, hipEventDisableTiming));
    M_HIP(hipEventCreate(&m->ev_t0));
    M_HIP(hipEventCreate(&m->ev_t1));
    if (m->transport == RCCL) {
        m->comm.assign(n, nullptr);
        M_NCCL(ncclCommInitAll(m->comm.data(), (int) n, devices));
    }
    return RT_OK;
}

extern "C" int rt_create_multi(rt_multi **out, const rt_scene_desc *scene, const int *devices, uint32_t n_devices, uint32_t band_rows, uint32_t parts,
                               uint32_t flags, uint32_t format)
{
    if (!out || !scene || !devices) return fail(RT_ERR_INVALID, "rt_create_multi: null argument");
    *out = nullptr;
    if (n_devices == 0 || n_devices > 64) return fail(RT_ERR_INVALID, "rt_create_multi: %u devices", n_devices);
    if (parts == 0) parts = 1;
    if (parts > 16) return fail(RT_ERR_INVALID, "rt_create_multi: %u parts per device (at most 16)", parts);
    if (flags & RT_FLAG_SIMPLE) return fail(RT_ERR_INVALID, "rt_create_multi: not available with RT_FLAG_SIMPLE");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RT_ERR_NO_DEVICE, "rt_create_multi: no HIP device available; this library has no CPU fallback");
    for (uint32_t r = 0; r < n_devices; r++)
        if (devices[r] < 0 || devices[r] >= ndev) return fail(RT_ERR_INVALID, "rt_create_multi: device %d out of range (%d devices)", devices[r], ndev);
    rt_multi *m = new (std::nothrow) rt_multi();
    if (!m) return fail(RT_ERR_NOMEM, "out of memory");
    DeviceRestore restore;
    int rc;
    try {
        rc = create_impl(m, scene, devices, n_devices, band_rows ? band_rows : 16, parts, flags, format);
    } catch (const std::bad_alloc &) {
        rc = fail(RT_ERR_NOMEM, "rt_create_multi: out of host memory");
    }
    if (rc != RT_OK) {
        std::string keep = rt_last_error();
        rt_multi_destroy(m);
        rt_set_last_error(keep.c_str());
        return rc;
    }
    *out = m;
    return RT_OK;
}

extern "C" int rt_render_multi(rt_multi *m, const double cam[16], void *root_full_fb, float *ms)
{
    if (!m || !cam) return fail(RT_ERR_INVALID, "rt_render_multi: null argument");
    if (m->in_flight_failed) return fail(RT_ERR_DEVICE, "rt_render_multi: an earlier frame of this object failed with part of it enqueued; destroy it and create a new one");
    DeviceRestore restore;
    const uint32_t n = m->n, P = m->parts;
    void *full = root_full_fb ? root_full_fb : m->full;
    struct FailGuard { // any early return between here and the end leaves sends / receives / events half issued
        rt_multi *m;
        bool ok = false;
        ~FailGuard() { if (!ok) m->in_flight_failed = true; }
    } guard{m};
    M_HIP(hipSetDevice(m->dev[0]));
    if (ms) M_HIP(hipEventRecord(m->ev_t0, m->s_render[0]));
    if (m->world == 1 && m->transport == DIRECT) { // one device, one part: the frame is this context's rows
        int rc = rt_render(m->ctx[0], cam, full, m->s_render[0], nullptr);
        if (rc != RT_OK) return rc;
    } else if (m->bandwise) {
        // Rows go band by band straight to where they belong in the full frame ("band-wise ncclRecv straight into final row
        // offsets"): band j of context q is rows (j W + q) B ... of the frame.  No rank-major receive slots and no reassembly pass over the
        // frame on the root.  Between distinct devices: one ncclSend / ncclRecv pair per band, all bands of a part in one group; rows that are
        // already on the root device (and every row in the one-GPU rehearsal) take ONE strided device copy per context.
        const size_t row_bytes = (size_t) m->width * m->pixel_bytes, band_bytes = row_bytes * m->band_rows;
        const size_t dst_pitch = band_bytes * m->world;
        for (uint32_t p = 0; p < P; p++) {
            for (uint32_t r = 0; r < n; r++) {
                const uint32_t q = p * n + r;
                M_HIP(hipSetDevice(m->dev[r]));
                M_HIP(hipStreamWaitEvent(m->s_render[r], m->ev_sent[q], 0)); // the previous frame's rows have left this buffer
                int rc = rt_render(m->ctx[q], cam, m->local[q], m->s_render[r], nullptr);
                if (rc != RT_OK) return rc;
                M_HIP(hipEventRecord(m->ev_rendered[q], m->s_render[r]));
                M_HIP(hipStreamWaitEvent(m->s_comm[r], m->ev_rendered[q], 0));
            }
            const bool by_rccl = m->transport == RCCL;
            if (by_rccl) M_NCCL(ncclGroupStart());
            ncclResult_t in_group = ncclSuccess;
            hipError_t copy_err = hipSuccess;
            for (uint32_t r = 0; r < n && in_group == ncclSuccess && copy_err == hipSuccess; r++) {
                const uint32_t q = p * n + r;
                const uint32_t full_bands = m->rows[q] / m->band_rows, tail_rows = m->rows[q] - full_bands * m->band_rows;
                char *dst0 = (char *) full + (size_t) q * band_bytes; // context q's band j starts dst_pitch * j further on
                if (by_rccl && (r != 0 || m->self_exchange)) {
                    for (uint32_t j = 0; j * m->band_rows < m->rows[q] && in_group == ncclSuccess; j++) {
                        const size_t bytes = j < full_bands ? band_bytes : (size_t) tail_rows * row_bytes;
                        in_group = ncclSend((const char *) m->local[q] + (size_t) j * band_bytes, bytes, ncclInt8, 0, m->comm[r], m->s_comm[r]);
                        if (in_group == ncclSuccess) in_group = ncclRecv(dst0 + (size_t) j * dst_pitch, bytes, ncclInt8, (int) r, m->comm[0], m->s_comm[0]);
                    }
                } else { // same device as the frame: one strided copy (and one more for a last, shorter band)
                    (void) hipSetDevice(m->dev[r]);
                    if (full_bands) copy_err = hipMemcpy2DAsync(dst0, dst_pitch, m->local[q], band_bytes, band_bytes, full_bands, hipMemcpyDeviceToDevice, m->s_comm[r]);
                    if (copy_err == hipSuccess && tail_rows)
                        copy_err = hipMemcpyAsync(dst0 + (size_t) full_bands * dst_pitch, (const char *) m->local[q] + (size_t) full_bands * band_bytes, (size_t) tail_rows * row_bytes,
                                                  hipMemcpyDeviceToDevice, m->s_comm[r]);
                }
            }
            if (by_rccl) {
                const ncclResult_t closed = ncclGroupEnd(); // (a failure inside the group still closes it before this call returns)
                if (in_group != ncclSuccess) return fail(RT_ERR_DEVICE, "ncclSend / ncclRecv of part %u failed: %s", p, ncclGetErrorString(in_group));
                M_NCCL(closed);
            }
            if (copy_err != hipSuccess) return fail(RT_ERR_DEVICE, "band copy of part %u failed: %s", p, hipGetErrorString(copy_err));
            for (uint32_t r = 0; r < n; r++) {
                const uint32_t q = p * n + r;
                M_HIP(hipSetDevice(m->dev[r]));
                M_HIP(hipEventRecord(m->ev_sent[q], m->s_comm[r]));
                if (r != 0 || !by_rccl) { // what ran on another comm stream than the root's: the root's comm stream waits for it
                    M_HIP(hipSetDevice(m->dev[0]));
                    M_HIP(hipStreamWaitEvent(m->s_comm[0], m->ev_sent[q], 0));
                }
            }
        }
        M_HIP(hipSetDevice(m->dev[0]));
        M_HIP(hipEventRecord(m->ev_gathered, m->s_comm[0]));
        M_HIP(hipStreamWaitEvent(m->s_render[0], m->ev_gathered, 0)); // the frame is complete for whatever follows on the root's render stream
    } else {
        // the root's receive slots are free again once the previous frame has been reassembled out of them
        if (m->have_assembled)
            for (uint32_t r = 0; r < n; r++) {
                M_HIP(hipSetDevice(m->dev[r]));
                if (m->transport != RCCL || r == 0) M_HIP(hipStreamWaitEvent(m->s_comm[r], m->ev_assembled, 0));
                if (r == 0 || m->transport == LOCAL_COPY) M_HIP(hipStreamWaitEvent(m->s_render[r], m->ev_assembled, 0));
            }
        for (uint32_t p = 0; p < P; p++) {
            // every device renders part p on its render stream; rows that stay on the root go straight into their slot
            for (uint32_t r = 0; r < n; r++) {
                const uint32_t q = p * n + r;
                M_HIP(hipSetDevice(m->dev[r]));
                void *dst = m->local[q] ? m->local[q] : (char *) m->gathered + (size_t) q * m->slot_bytes;
                if (m->local[q]) M_HIP(hipStreamWaitEvent(m->s_render[r], m->ev_sent[q], 0)); // the previous frame's rows have left this buffer
                int rc = rt_render(m->ctx[q], cam, dst, m->s_render[r], nullptr);
                if (rc != RT_OK) return rc;
                M_HIP(hipEventRecord(m->ev_rendered[q], m->s_render[r]));
                M_HIP(hipStreamWaitEvent(m->s_comm[r], m->ev_rendered[q], 0));
            }
            // ... and part p travels on the comm streams while part p + 1 renders
            if (m->transport == RCCL) {
                M_NCCL(ncclGroupStart());
                ncclResult_t in_group = ncclSuccess; // a failure inside the group still closes it before this call returns
                for (uint32_t r = 0; r < n && in_group == ncclSuccess; r++) {
                    const uint32_t q = p * n + r;
                    if (!m->local[q]) continue;
                    in_group = ncclSend(m->local[q], m->slot_bytes, ncclInt8, 0, m->comm[r], m->s_comm[r]);
                    if (in_group == ncclSuccess)
                        in_group = ncclRecv((char *) m->gathered + (size_t) q * m->slot_bytes, m->slot_bytes, ncclInt8, (int) r, m->comm[0], m->s_comm[0]);
                }
                const ncclResult_t closed = ncclGroupEnd();
                if (in_group != ncclSuccess) return fail(RT_ERR_DEVICE, "ncclSend / ncclRecv of part %u failed: %s", p, ncclGetErrorString(in_group));
                M_NCCL(closed);
            } else if (m->transport == LOCAL_COPY) {
                for (uint32_t r = 1; r < n; r++) {
                    const uint32_t q = p * n + r;
                    M_HIP(hipMemcpyAsync((char *) m->gathered + (size_t) q * m->slot_bytes, m->local[q], m->slot_bytes, hipMemcpyDeviceToDevice, m->s_comm[r]));
                }
            }
            for (uint32_t r = 0; r < n; r++) {
                const uint32_t q = p * n + r;
                if (!m->local[q]) continue;
                M_HIP(hipSetDevice(m->dev[r]));
                M_HIP(hipEventRecord(m->ev_sent[q], m->s_comm[r]));
                if (m->transport == LOCAL_COPY) { // the copy ran on the sender's comm stream: the root's comm stream waits for it
                    M_HIP(hipSetDevice(m->dev[0]));
                    M_HIP(hipStreamWaitEvent(m->s_comm[0], m->ev_sent[q], 0));
                }
            }
        }
        // root: everything has arrived on its comm stream -> reassemble on its render stream
        M_HIP(hipSetDevice(m->dev[0]));
        M_HIP(hipEventRecord(m->ev_gathered, m->s_comm[0]));
        M_HIP(hipStreamWaitEvent(m->s_render[0], m->ev_gathered, 0));
        int rc = rt_assemble(m->ctx[0], m->gathered, full, m->s_render[0]);
        if (rc != RT_OK) return rc;
        M_HIP(hipEventRecord(m->ev_assembled, m->s_render[0]));
        m->have_assembled = true;
    }
    if (ms) {
        M_HIP(hipEventRecord(m->ev_t1, m->s_render[0]));
        M_HIP(hipEventSynchronize(m->ev_t1));
        M_HIP(hipEventElapsedTime(ms, m->ev_t0, m->ev_t1));
    }
    m->last_full = full;
    guard.ok = true;
    return RT_OK;
}

extern "C" int rt_multi_wait(rt_multi *m)
{
    if (!m) return fail(RT_ERR_INVALID, "rt_multi_wait: null argument");
    DeviceRestore restore;
    M_HIP(hipSetDevice(m->dev[0]));
    M_HIP(hipStreamSynchronize(m->s_render[0]));
    return RT_OK;
}

extern "C" void *rt_multi_fb(rt_multi *m) { return m ? m->full : nullptr; }

extern "C" void *rt_multi_stream(rt_multi *m) { return m ? (void *) m->s_render[0] : nullptr; }

extern "C" int rt_multi_download(rt_multi *m, void *host_dst, size_t bytes)
{
    if (!m || !host_dst) return fail(RT_ERR_INVALID, "rt_multi_download: null argument");
    if (bytes > m->full_bytes) return fail(RT_ERR_INVALID, "rt_multi_download: %zu bytes requested, the frame holds %zu", bytes, m->full_bytes);
    DeviceRestore restore;
    M_HIP(hipSetDevice(m->dev[0]));
    if (!m->last_full) return fail(RT_ERR_INVALID, "rt_multi_download: no frame has been rendered yet");
    M_HIP(hipStreamSynchronize(m->s_render[0]));
    M_HIP(hipMemcpy(host_dst, m->last_full, bytes, hipMemcpyDeviceToHost)); // (the caller's own buffer when the last frame was rendered into one)
    return RT_OK;
}

extern "C" int rt_multi_info(const rt_multi *m, uint32_t *n_contexts, uint32_t *transport)
{
    if (!m) return fail(RT_ERR_INVALID, "rt_multi_info: null argument");
    if (n_contexts) *n_contexts = m->world;
    if (transport) *transport = (uint32_t) m->transport;
    return RT_OK;
}
