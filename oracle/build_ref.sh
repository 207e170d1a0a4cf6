#!/bin/bash
# Builds oracle/_ref/ from the reference sources WHERE THEY LIE under /root/reference.
# TEST INFRASTRUCTURE: outputs only into oracle/_ref/ (git-ignored, but it travels to the GPU box).
#   ref_host_dump_O0 : the reference's host classes (Model, AccelerationStructureExplicit,
#       Camera, Resource) + oracle/ref_host_dump.cpp, compiled with g++ directly (no cmake),
#       unoptimised like the reference CI (no CMAKE_BUILD_TYPE).
#   <kernel>.strict.co / <kernel>.default.co : the reference's own OpenCL kernels compiled for
#       gfx950 by the image's ROCm clang with the image's real OpenCL device libraries -- what
#       clBuildProgram would produce on the MI355X.  "strict" = -ffp-contract=off
#       -cl-fp32-correctly-rounded-divide-sqrt (the floating-point model oracle/lt_oracle.c
#       restates); "default" = what NULL build options give (renderer_opencl.cpp:50).
# No-op (exit 0) when /root/reference is absent (the GPU box uses the prebuilt files).
set -e
REF=${LT_REFERENCE_DIR:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
if [ ! -d "$REF/src" ]; then echo "build_ref: $REF not present, keeping prebuilt oracle/_ref"; exit 0; fi
mkdir -p "$OUT"
CLANG=/opt/rocm/lib/llvm/bin/clang
for opt in O0; do
  g++ -std=c++17 -$opt -w -I"$REF/include" "$HERE/ref_host_dump.cpp" \
      "$REF/src/model.cpp" "$REF/src/acceleration_structure_explicit.cpp" "$REF/src/camera.cpp" "$REF/src/resource.cpp" \
      -o "$OUT/ref_host_dump_$opt"
done
declare -A K=(
  [basic]=resources/kernels/opencl/basic.cl
  [basic_lighting]=resources/kernels/opencl/basic_lighting.cl
  [accumulator]=examples/accumulator/resources/kernels/accumulator.cl
  [global_illumination]=examples/global_illumination/resources/kernels/global_illumination.cl
  [global_illumination25]=resources/kernels/opencl/global_illumination.cl
  [custom_opencl]=examples/custom_kernel/resources/kernels/custom_opencl.cl
)
for name in "${!K[@]}"; do
  src="$REF/${K[$name]}"
  $CLANG -x cl -cl-std=CL2.0 -target amdgcn-amd-amdhsa -mcpu=gfx950 -Xclang -finclude-default-header -O3 -w \
      -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt "$src" -o "$OUT/$name.strict.co"
  $CLANG -x cl -cl-std=CL2.0 -target amdgcn-amd-amdhsa -mcpu=gfx950 -Xclang -finclude-default-header -O3 -w \
      "$src" -o "$OUT/$name.default.co"
done
ls -la "$OUT"
