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
}  // namespace dust_internal
