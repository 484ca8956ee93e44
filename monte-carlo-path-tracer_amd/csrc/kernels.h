// Launchers implemented in kernels.hip (device code) and called from mcpt_api.cpp (host, C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "device_scene.h"

hipError_t launch_render(const DevScene& sc, const RenderParams& p, float4* accum, DevCounters* cnt, hipStream_t stream);
hipError_t launch_probe_paths(const DevScene& sc, const RenderParams& p, uint32_t n, const double* o, const double* d, float* out3, DevCounters* cnt, hipStream_t stream);
hipError_t launch_tonemap(const float4* accum, uint8_t* rgb, int w, int h, int flip, hipStream_t stream);
hipError_t launch_probe_trace(const DevScene& sc, uint32_t n, const double* o, const double* d, const double* t1, const double* t2, int any_hit,
                              float* out_t, int* out_tri, float* out_u, float* out_v, hipStream_t stream);
hipError_t launch_probe_cast_ray(const DevScene& sc, uint32_t n, const int* xy, const float* xi, float* out6, hipStream_t stream);
hipError_t launch_probe_bsdf(uint32_t n, const float* normal, const float* wi, const float* kd, const float* ks, const float* ns, const float* wo,
                             const float* xi, float* out12, hipStream_t stream);
hipError_t launch_probe_hit_shade(const DevScene& sc, uint32_t n, const int* tri, const float* u, const float* v, const double* dir, float* out6, hipStream_t stream);
hipError_t launch_probe_sample_light(const DevScene& sc, uint32_t n, const double* point, const float* xi, float* out10, hipStream_t stream);
hipError_t launch_probe_texture(const DevScene& sc, int material, uint32_t n, const float* uv, float* out3, hipStream_t stream);
hipError_t launch_probe_rng(uint32_t n, const uint32_t* key3, uint32_t seed_lo, uint32_t seed_hi, float* out4, hipStream_t stream);
