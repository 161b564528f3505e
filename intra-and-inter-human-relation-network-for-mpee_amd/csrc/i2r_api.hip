// C-ABI plumbing: error string, device check, and the program runner that replays a list of launches.
#include <stdarg.h>
#include <string.h>

#include "i2r_common.h"

static thread_local char g_err[512] = "";
thread_local hipEvent_t i2r_tls_t0 = nullptr, i2r_tls_t1 = nullptr;  // timing pair of the op being replayed (i2r_common.h: i2r_launch)

void i2r_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* i2r_last_error(void) { return g_err; }
extern "C" int i2r_abi_version(void) { return I2R_ABI_VERSION; }

extern "C" int i2r_device_check(int32_t dev, int32_t* cu_count, int32_t* lds_bytes) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        i2r_set_error("i2r_device_check: no HIP device %d", dev);
        return I2R_E_NODEV;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int32_t)prop.sharedMemPerBlock;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        i2r_set_error("i2r_device_check: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        return I2R_E_NODEV;
    }
    return I2R_OK;
}

// ---- device-side lane synchronisation (round 6) -------------------------------------------------------------------------------------
// A cross-stream dependency through events (hipEventRecord -> hipStreamWaitEvent) costs the WAITING stream ~10 us after the producer
// ends, ~20 us inside a forward (marker packet + cross-queue barrier packet; tools/probe/xstream_latency*.hip), and an HRFormer forward
// crosses lanes on its critical path 13 times.  Behind an I2R_OP_LANE_FLAGS op (a flag buffer) the same ops are two one-wave kernels: the
// producer's stream sets flags behind its work (lane_signal_k), the consumer's stream runs lane_wait_k, which spins (with s_sleep)
// until its flag is set, clears it and ends -- the consumer's next kernel follows in stream order.  Kernel boundaries give the same
// release / acquire as with events.  Preconditions (the CALLER checks them, engine.Program.run): every lane is its own hardware queue
// (a waiting kernel must never sit in front of the kernel that signals it) and the program issues every signal before the waits for it
// (launch-list order = host enqueue order).  A wait gives up after ~50 ms and raises flags[I2R_FLAG_TIMEOUT] instead of hanging the GPU.
constexpr int I2R_FLAG_TIMEOUT = 63;
__global__ void lane_signal_k(int* flags, int row, int mask) {
    const int l = threadIdx.x;
    if (l < 4 && (mask & (1 << l))) __hip_atomic_store(flags + row * 4 + l, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void lane_wait_k(int* flags, int idx) {
    if (threadIdx.x != 0) return;
    long long t0 = wall_clock64();
    while (__hip_atomic_load(flags + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 5000000LL) {  // 50 ms at 100 MHz
            __hip_atomic_store(flags + I2R_FLAG_TIMEOUT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    __hip_atomic_store(flags + idx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static inline void lane_signal(int* flags, int row, int mask, void* stream) {
    lane_signal_k<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(flags, row, mask);
}
static inline void lane_wait(int* flags, int row, int lane, void* stream) {
    lane_wait_k<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>(flags, row * 4 + lane);
}

// t0 / t1 (both or neither): per-op timing events of i2r_run_program_timed, recorded on the op's own stream around its launch
static int run_program(const i2r_op* ops, int32_t n_ops, void* const* streams, void* const* events, void* const* t0, void* const* t1) {
    I2R_CHECK_ARG(ops && n_ops >= 0, "i2r_run_program: null program");
    int next_event = 0;
    void* slot_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // stream of the last RECORD per slot
    bool slot_set[8] = {false, false, false, false, false, false, false, false};                      // (the null stream is a valid stream)
    // flag rows: 0..7 = record slots, 8 = fork, 8 + l = join from lane l (l >= 1): [row][consumer lane]
    int* flags = nullptr;  // set by I2R_OP_LANE_FLAGS: the device-side form of the sync ops behind it
    for (int i = 0; i < n_ops; ++i) {
        const i2r_op& op = ops[i];
        int rc = I2R_OK;
        if (op.kind == I2R_OP_LANE_FLAGS) {
            flags = streams ? (int*)op.args : nullptr;
            continue;
        }
        if (op.kind == I2R_OP_XSYNC) {  // all-to-all among the lanes of the mask: one event per lane, every other lane waits for it
            if (!streams) continue;  // single-stream replay: program order already is the order
            bool distinct = false;
            for (int l = 1; l < 4; ++l)
                if ((op.lane & (1 << l)) && streams[l] != streams[0]) distinct = true;
            if (!distinct) continue;
            I2R_CHECK_ARG(events, "i2r_run_program: xsync needs events");
            hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int l = 0; l < 4 && rc == I2R_OK; ++l)
                if (op.lane & (1 << l)) {
                    ev[l] = (hipEvent_t)events[next_event++ & 7];
                    if (hipEventRecord(ev[l], (hipStream_t)streams[l]) != hipSuccess) rc = I2R_E_LAUNCH;
                }
            for (int d = 0; d < 4 && rc == I2R_OK; ++d)
                if (op.lane & (1 << d))
                    for (int l = 0; l < 4 && rc == I2R_OK; ++l)
                        if (l != d && ev[l] && streams[l] != streams[d])
                            if (hipStreamWaitEvent((hipStream_t)streams[d], ev[l], 0) != hipSuccess) rc = I2R_E_LAUNCH;
            if (rc != I2R_OK) {
                i2r_set_error("i2r_run_program: xsync failed at op %d", i);
                return rc;
            }
            continue;
        }
        if (op.kind == I2R_OP_RECORD || op.kind == I2R_OP_WAIT) {  // point-to-point: lane = op.lane & 3, event 8 + slot
            if (!streams) continue;
            const int l = op.lane & 3, slot = (op.lane >> 8) & 7, consumers = (op.lane >> 16) & 15;
            I2R_CHECK_ARG(events, "i2r_run_program: record / wait needs events");
            hipEvent_t ev = (hipEvent_t)events[8 + slot];
            if (op.kind == I2R_OP_RECORD) {
                slot_stream[slot] = streams[l];
                slot_set[slot] = true;
                if (flags) lane_signal(flags, slot, consumers & ~(1 << l), streams[l]);
                else if (hipEventRecord(ev, (hipStream_t)streams[l]) != hipSuccess) rc = I2R_E_LAUNCH;
            } else {
                I2R_CHECK_ARG(slot_set[slot], "i2r_run_program: op %d waits for slot %d before any record", i, slot);
                if (flags) lane_wait(flags, slot, l, streams[l]);
                else if (slot_stream[slot] != streams[l] && hipStreamWaitEvent((hipStream_t)streams[l], ev, 0) != hipSuccess) rc = I2R_E_LAUNCH;
            }
            if (flags && hipGetLastError() != hipSuccess) rc = I2R_E_LAUNCH;
            if (rc != I2R_OK) {
                i2r_set_error("i2r_run_program: record / wait failed at op %d", i);
                return rc;
            }
            continue;
        }
        if (op.kind == I2R_OP_FORK || op.kind == I2R_OP_JOIN) {
            I2R_CHECK_ARG(streams && events, "i2r_run_program: fork/join needs streams and events");
            if (flags) {  // (device-side form, see lane_signal_k)
                if (op.kind == I2R_OP_FORK) {
                    lane_signal(flags, 8, op.lane & 14, streams[0]);
                    for (int l = 1; l < 4; ++l)
                        if (op.lane & (1 << l)) lane_wait(flags, 8, l, streams[l]);
                } else {
                    for (int l = 1; l < 4; ++l)
                        if (op.lane & (1 << l)) {
                            lane_signal(flags, 8 + l, 1, streams[l]);
                            lane_wait(flags, 8 + l, 0, streams[0]);
                        }
                }
                if (hipGetLastError() != hipSuccess) {
                    i2r_set_error("i2r_run_program: device-side fork/join failed at op %d", i);
                    return I2R_E_LAUNCH;
                }
                continue;
            }
            if (op.kind == I2R_OP_FORK) {  // lanes in the mask wait for everything issued on lane 0 so far
                hipEvent_t ev = (hipEvent_t)events[next_event++ & 7];
                if (hipEventRecord(ev, (hipStream_t)streams[0]) != hipSuccess) rc = I2R_E_LAUNCH;
                for (int l = 1; l < 4 && rc == I2R_OK; ++l)
                    if (op.lane & (1 << l))
                        if (hipStreamWaitEvent((hipStream_t)streams[l], ev, 0) != hipSuccess) rc = I2R_E_LAUNCH;
            } else {  // lane 0 waits for the lanes in the mask
                for (int l = 1; l < 4 && rc == I2R_OK; ++l)
                    if (op.lane & (1 << l)) {
                        hipEvent_t ev = (hipEvent_t)events[next_event++ & 7];
                        if (hipEventRecord(ev, (hipStream_t)streams[l]) != hipSuccess ||
                            hipStreamWaitEvent((hipStream_t)streams[0], ev, 0) != hipSuccess)
                            rc = I2R_E_LAUNCH;
                    }
            }
            if (rc != I2R_OK) {
                i2r_set_error("i2r_run_program: event fork/join failed at op %d", i);
                return rc;
            }
            continue;
        }
        I2R_CHECK_ARG(op.lane >= 0 && op.lane < 4 && op.args, "i2r_run_program: op %d bad lane/args", i);
        void* st = streams ? streams[op.lane] : nullptr;
        I2R_CHECK_ARG(streams || op.lane == 0, "i2r_run_program: op %d uses lane %d without streams", i, op.lane);
        if (t0 && t1 && t1[i]) {
            i2r_tls_t0 = (hipEvent_t)t0[i];
            i2r_tls_t1 = (hipEvent_t)t1[i];
        }
        switch (op.kind) {
            case I2R_OP_CONV: rc = i2r_conv((const i2r_conv_desc*)op.args, st); break;
            case I2R_OP_CONV_GROUP: {
                const i2r_conv_group_args* a = (const i2r_conv_group_args*)op.args;
                rc = i2r_conv_grouped(a->d, a->n, a->block_map, a->map_len, st);
                break;
            }
            case I2R_OP_STEM: {
                const i2r_stem_args* a = (const i2r_stem_args*)op.args;
                rc = i2r_stem_conv(a->in, a->w, a->bias, a->out, a->n_img, a->cin, a->in_h, a->in_w, a->cout, a->out_cs, a->n_src, a->n_valid, a->out_dt, st);
                break;
            }
            case I2R_OP_PE_RES_STEM: {
                const i2r_pe_res_args* a = (const i2r_pe_res_args*)op.args;
                rc = i2r_pe_res_stem(a->in, a->w_pre, a->w7, a->bias, a->out, a->n_img, a->in_h, a->in_w, a->cout, a->out_cs, a->n_src, a->n_valid, st);
                break;
            }
            case I2R_OP_MAXPOOL: {
                const i2r_pool_args* a = (const i2r_pool_args*)op.args;
                rc = i2r_maxpool3x3s2(a->in, a->out, a->n_img, a->in_h, a->in_w, a->c, a->in_cs, a->out_cs, st);
                break;
            }
            case I2R_OP_HEAD: {
                const i2r_head_args* a = (const i2r_head_args*)op.args;
                rc = i2r_head(a->in, a->w, a->bias, a->out, a->n_img, a->h, a->w_, a->cin, a->in_cs, a->cout, st);
                break;
            }
            case I2R_OP_LAYERNORM: {
                const i2r_ln_args* a = (const i2r_ln_args*)op.args;
                rc = i2r_layernorm(a->in, a->w, a->b, a->out, a->npix, a->c, a->cs, a->eps, a->out_dt, st);
                break;
            }
            case I2R_OP_CONV_CHAIN:
                rc = i2r_conv_chain((const i2r_conv_chain_args*)op.args, st);
                break;
            case I2R_OP_WINATTN: {
                const i2r_winattn_args* a = (const i2r_winattn_args*)op.args;
                rc = i2r_window_attn(a->qkv, a->bias, a->out, a->n_img, a->h, a->w_, a->c, a->cs, a->heads, st);
                break;
            }
            case I2R_OP_HRT_ATTN: {
                const i2r_hrt_attn_args* a = (const i2r_hrt_attn_args*)op.args;
                rc = i2r_hrt_attn_block(a->x, a->out, a->ln_w, a->ln_b, a->wqkv, a->bqkv, a->wo, a->bo, a->n_img, a->h, a->w_, a->c, a->cs, a->heads,
                                        a->eps, a->dtype, a->variant, st);
                break;
            }
            case I2R_OP_HRT_MLP: {
                const i2r_hrt_mlp_args* a = (const i2r_hrt_mlp_args*)op.args;
                rc = i2r_hrt_mlp_block(a->x, a->out, a->ln_w, a->ln_b, a->w1, a->b1, a->wdw, a->bdw, a->w2, a->b2, a->n_img, a->h, a->w_, a->c, a->cs,
                                       a->hidden_pad, a->eps, a->dtype, a->variant, st);
                break;
            }
            case I2R_OP_DWCONV: {
                const i2r_dw_args* a = (const i2r_dw_args*)op.args;
                rc = i2r_dwconv3x3(a->in, a->w, a->bias, a->out, a->n_img, a->in_h, a->in_w, a->c, a->cs, a->stride, a->act, a->dt, st);
                break;
            }
            case I2R_OP_UPSAMPLE: rc = i2r_upsample_bilinear_add_multi((const i2r_up_args*)op.args, st); break;
            case I2R_OP_FUSE_UP: {
                const i2r_fuse_up_args* a = (const i2r_fuse_up_args*)op.args;
                rc = i2r_fuse_up_add(a->base, a->t1, a->s1, a->t2, a->s2, a->out, a->n_img, a->h, a->w, a->cs, a->act, a->dt, st);
                break;
            }
            case I2R_OP_CONV1X1_PAIR: rc = i2r_conv1x1_pair((const i2r_conv1x1_pair_args*)op.args, st); break;
            case I2R_OP_CONV1X1_LP: rc = i2r_conv1x1_lp((const i2r_conv1x1_lp_args*)op.args, st); break;
            case I2R_OP_ROWS_GATHER: {
                const i2r_gather_args* a = (const i2r_gather_args*)op.args;
                rc = i2r_rows_gather(a->src, a->out, a->map, a->n_out, a->floats_per_crop, st);
                break;
            }
            case I2R_OP_VIEW_SCRAMBLE: {
                const i2r_scramble_args* a = (const i2r_scramble_args*)op.args;
                rc = i2r_view_scramble(a->o, a->out, a->person_map, a->n_out, a->n_images, a->max_persons, a->c, a->cs, a->hw, st);
                break;
            }
            case I2R_OP_PE_CAT_VEC: rc = i2r_pe_cat_vec((const i2r_pe_cat_vec_args*)op.args, st); break;
            case I2R_OP_MH_ATTN: rc = i2r_mh_attention((const i2r_mh_attn_args*)op.args, st); break;
            case I2R_OP_ENC_KV: rc = i2r_encoder_kv((const i2r_encoder_desc*)op.args, st); break;
            case I2R_OP_ENC_LAYER: rc = i2r_encoder_layer((const i2r_encoder_desc*)op.args, st); break;
            default: i2r_set_error("i2r_run_program: op %d unknown kind %d", i, op.kind); return I2R_E_ARG;
        }
        i2r_tls_t0 = i2r_tls_t1 = nullptr;
        if (rc != I2R_OK) return rc;
    }
    return I2R_OK;
}

extern "C" int i2r_run_program(const i2r_op* ops, int32_t n_ops, void* const* streams, void* const* events) {
    return run_program(ops, n_ops, streams, events, nullptr, nullptr);
}

extern "C" int i2r_run_program_timed(const i2r_op* ops, int32_t n_ops, void* const* streams, void* const* events, void* const* t0,
                                     void* const* t1) {
    I2R_CHECK_ARG(t0 && t1, "i2r_run_program_timed: null timing event arrays");
    return run_program(ops, n_ops, streams, events, t0, t1);
}
