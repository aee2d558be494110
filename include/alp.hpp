// include/alp.hpp — source-compatible drop-in for the reference's umbrella header (cwida/ALP include/alp.hpp:4-13).
//
// Same namespaces, types and static functions as the reference's vector API, but every function forwards to the
// MI355X kernels in libalpgpu.so through the C ABI (include/alpgpu.h): there is no CPU implementation behind this
// header.  Each call ships ONE 1024-value vector (or one rowgroup for init) to the GPU and back, so it is a
// compatibility surface — code that cares about throughput calls the batch entry points of include/alpgpu.h
// (alpgpu_encode_f64 / alpgpu_decode_f64); code that keeps the reference's per-vector loop shape calls alp::gpu::rowgroup<PT>
// (alp/batch.hpp) once per rowgroup instead of four functions per vector.
// Link with -lalpgpu.  See INTEGRATION.md.
#ifndef ALP_HPP
#define ALP_HPP

#include "alp/config.hpp"
#include "alp/constants.hpp"
#include "alp/common.hpp"
#include "alp/gpu_bridge.hpp"
#include "alp/decoder.hpp"
#include "alp/sampler.hpp"
#include "alp/encoder.hpp"
#include "alp/falp.hpp"
#include "alp/rd.hpp"
#include "alp/batch.hpp"
#include "fastlanes/ffor.hpp"
#include "fastlanes/unffor.hpp"

#endif // ALP_HPP
