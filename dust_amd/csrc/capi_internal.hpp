// capi_internal.hpp -- what comm.hip needs of the handles capi.cpp defines (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/dust_hip.h"

namespace dust_internal {
DustStatus set_error(DustStatus status, const std::string& message);  // dust_hip_last_error() of the calling thread
hipStream_t context_stream(DustHipContext*);
int context_device(DustHipContext*);
void context_retain(DustHipContext*);
void context_release(DustHipContext*);
DustHipContext* pipeline_context(DustHipPipeline*);
void pipeline_size(DustHipPipeline*, uint32_t* width, uint32_t* height);
// the pipeline takes part in a collective of a communicator with world > 1: DUST_RESERVE_AUTO leaves workgroup slots free from now on
void pipeline_note_collective(DustHipPipeline*);
// a pending sharded surfel trace (DustHipFrameParams::surfel_world): its slot-ordered staging arrays {32, 16, 16 bytes per slot}, and the pass's second half
struct SurfelStage { void* req; void* repl; void* sun; uint32_t slots_per_rank, rank, pool_size; };
DustStatus surfel_stage_view(DustHipPipeline*, uint32_t world, SurfelStage* out);
DustStatus surfel_finish(DustHipPipeline*, uint32_t frame_index);
// the exchange buffers dust_hip_pipeline_gi_exchange(p, padded_rows) made, WITHOUT (re)making them: DUST_ERR_NOT_READY when the
// pipeline's buffers were prepared for another padded_rows (or not at all) -- a frame's stamps must not be dropped by a re-allocation
DustStatus gi_exchange_view(DustHipPipeline*, uint32_t padded_rows, DustHipGiExchange* out);
// a stream of another object (a communicator's) that reads the context's pipelines: sync_stream waits for it too
void context_add_stream(DustHipContext*, hipStream_t);
void context_remove_stream(DustHipContext*, hipStream_t);
}  // namespace dust_internal
