// include/alp.hpp — source-compatible drop-in for the reference's umbrella header (cwida/ALP include/alp.hpp:4-13).
// READ THIS FIRST: the per-vector functions below are CORRECT but SLOW — one PCIe round trip per 1024 values, 4-10 k vectors/s (33-80 MB/s), about a
// hundred times slower than the reference's CPU loop.  Replacing the include path and the link line gives parity, not speed; speed is
// alp::gpu::rowgroup<PT> / alp::gpu::column<PT> (alp/batch.hpp: 30-40 GB/s host to host) or the C ABI on device-resident columns (TB/s).  INTEGRATION.md §1.
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
