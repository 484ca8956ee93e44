// Wavefront pipeline state: a pool of path SLOTS in HBM (structure of arrays, 16-B records, one slot per thread of the
// shade kernel => fully coalesced), the shadow-ray queue, and the tiny control block the kernels use to hand over counts
// without a host round trip.  See DESIGN.md §Kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_scene.h"

#ifndef WF_SHADE_BLOCK
#define WF_SHADE_BLOCK 256        // threads (= pool slots) per shade workgroup: the unit of the class sort, of a shadow-queue region and of a trace chunk
#endif
#define SLOT_DEAD  0u   // no work item bound
#define SLOT_ALIVE 1u   // a path is in flight; its extend ray was (or is about to be) traced
#define SLOT_DRAIN 2u   // path ended in the same shade call that emitted its last shadow ray: finalise next call

struct PathPool {
    float4* ray_o;      // extend-ray origin xyz (= previous path vertex) | w: int bits = light triangle the slot's shadow ray skips
    float4* ray_d;      // extend-ray direction xyz                        | w: uint bits, bit 0 = an extend ray is pending this iteration
    float4* hit;        // written by trace: int bits tri (leaf order) | lobe class << 28 (-1 = miss), u, v, t
    float4* sq_o;       // shadow-ray records, indexed like shadow_queue: origin xyz | w: int bits = the sampled light triangle to skip
    float4* sq_d;       //                                                direction xyz | w: t2
    float4* nee;        // radiance the pending shadow ray carries if unoccluded (xyz) | w: uint, set non-zero by trace if the ray is
                        // blocked; otherwise xyz is added to L by the NEXT shade call
    float4* L;          // radiance of the current path so far xyz | w: pdf of the BSDF sample that produced the extend ray
    float4* beta;       // path throughput xyz | w: uint bits  state(2) | prev_mirror(1) | shadow ray pending(1) | bounce << 8
    float4* sum;        // item accumulator: sum of finished samples xyz | w: number of finished samples
    uint4* ids;         // 16 B per slot, used as two arrays of 8 B: [P x {pixel, current sample index}] [P x {next sample index, end sample index}]
    uint32_t* shadow_queue;   // slots with a pending shadow ray: shade block b owns entries [256 b, 256 b + shadow_count[b]) -- no atomics
    uint32_t* shadow_count;   // entries each shade block wrote this iteration
    uint32_t* live_cnt;       // per shade block: slots that are not DEAD after this iteration's shade call (input of the drain compaction)
    uint2* block_items;       // per shade block: {next, end} of its private work-item range (RenderParams::priv_items); the block alone reads and writes it
    uint32_t P;         // slots
};

#define WF_SHARDS 8          // shadow-queue shards (block b appends to shard b % 8: 8x less contention on the cursor)
#define WF_COUNTER_REPLICAS 1024
#define WF_ITEM_SHARDS 64
#define WF_LDS_MATS 16        // material / light tables up to these sizes are staged in LDS by the shade kernel
#define WF_LDS_LIGHTS 8

struct IterCtl {        // indexed [iteration & 3]; shade(it) zeroes entry (it+1)&3 for the next iteration
    uint32_t trace_head[4];
    uint32_t any_active[4];              // set (plain store) by any wave that still owns a live slot
    uint32_t pad[8 + 4 * WF_SHARDS];     // [0] trace watchdog flag; drain compaction (wf_compact_*): [1] slots the kernels sweep (0 = all of the pool), [2] "compact now",
                                         // [3] slots after this compaction, [4] live slots found, [5] compactions done in this job
    struct { uint32_t v; uint32_t pad[15]; } item_cursor[WF_ITEM_SHARDS];   // work-item cursors, one 64-B line each
};

// Work items are handed out in units of WF_SHADE_BLOCK consecutive items; unit u belongs to shard u % WF_ITEM_SHARDS.
// local index l of shard k  ->  global item ((l / WF_SHADE_BLOCK) * WF_ITEM_SHARDS + k) * WF_SHADE_BLOCK + l % WF_SHADE_BLOCK
__host__ __device__ inline uint32_t wf_shard_capacity(uint32_t n_items, uint32_t k) {
    const uint32_t units = (n_items + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK;
    return units > k ? ((units - k + WF_ITEM_SHARDS - 1) / WF_ITEM_SHARDS) * WF_SHADE_BLOCK : 0u;
}

#define WF_CTL_WATCHDOG 0
#define WF_CTL_P_ACTIVE 1
#define WF_CTL_DO_COMPACT 2
#define WF_CTL_P_NEXT 3
#define WF_CTL_LIVE 4
#define WF_CTL_COMPACTIONS 5
// End-of-job drain: once the work items have run out the slots die one by one, all over the pool, and the kernels keep sweeping a pool that is mostly
// dead (a 1024-spp job spends its last ~10 of 384 iterations like that; the 128-spp share of an 8-way strong-scaled split 10 of 58).  When at most half
// of the swept slots are alive the live ones are moved to the front of the pool (through a scratch copy: wf_compact_move / _back) and the sweep shrinks
// to them (IterCtl::pad[WF_CTL_P_ACTIVE]).  A slot's number means nothing to the path it holds -- pixel and sample index travel in its `ids` record --
// so the film does not change.
struct CompactBufs {
    float4 *beta, *L, *ray_d, *ray_o, *hit, *nee; uint2* ids;   // scratch for pool.P / 2 + 4096 slots
    uint32_t* dst_off;                                          // per shade block: where its live slots go (exclusive prefix of live_cnt)
    uint32_t capacity;                                          // slots the scratch holds
    uint32_t eighths;                                           // compact when at most eighths / 8 of the swept slots are alive
};
#define WF_COMPACT_MIN_SLOTS (1u << 18)                         // pools sweeping fewer slots than this are left alone

struct WaveTuning {     // scheduler thresholds of the trace kernel (lanes out of 64)
    uint32_t refill_at;       // refill when at least this many lanes are idle
    uint32_t leaf_at;         // run the leaf block when at least this many lanes wait at a leaf
    uint32_t inner_keep;      // keep iterating the inner-node block while at least this many lanes are at inner nodes
    uint32_t policy;          // 0 = fixed thresholds above; 1 = greedy: run the block most lanes can take part in
    uint32_t pend_cap;        // run the leaf block at the latest when this many lanes carry a parked leaf (speculative traversal)
};

hipError_t launch_wf_shade(const DevScene& sc, const RenderParams& p, const PathPool& pool, IterCtl* ctl, uint32_t iteration, uint32_t n_items,
                           float4* accum, DevCounters* cnt, hipStream_t stream);
hipError_t launch_wf_trace(const DevScene& sc, const PathPool& pool, IterCtl* ctl, uint32_t iteration, const WaveTuning& tune, bool count,
                           DevCounters* cnt, uint32_t grid_blocks, int* stack_overflow, hipStream_t stream);
hipError_t launch_wf_pool_reset(const PathPool& pool, IterCtl* ctl, hipStream_t stream);   // job start: every slot DEAD, control block zeroed
hipError_t launch_wf_compact(const PathPool& pool, const CompactBufs& cb, IterCtl* ctl, uint32_t iteration, uint32_t n_shared, uint32_t priv_items, hipStream_t stream);
int wf_trace_blocks_per_cu(bool count);
uint32_t wf_trace_block_threads();
void wf_make_fastdiv(uint32_t d, uint32_t& m, uint32_t& s);   // exact x / d == (uint64(x) * m) >> s for x < MCPT_FASTDIV_MAX (RenderParams::div_*)
size_t wf_trace_overflow_bytes_per_lane(uint32_t wide_depth);
