#!/bin/bash
# oracle/ref_recipe/build_ref.sh -- compiles the reference's OWN hot-path sources, from where they lie under $KHRONOS_ROOT
# (default /root/reference; nothing is copied), against the functional stand-ins of oracle/ref_recipe/standin, together with
# oracle/ref_recipe/ref_harness.cpp, into oracle/_ref/libref_khronos.so (git-ignored; it travels to the GPU box with the
# snapshot like our own built .so files).  tests/test_cpu_ref_pin.py runs it beside the oracle.
#
# This is NOT the reference's build (that needs Hydra, spatial_hash, config_utilities, spark_dsg, Eigen, OpenCV, glog: none
# is in the image).  What it executes is the logic of the files below; the containers are ours (ref_standin.h says which
# ASSUMPTIONS.md item each one stands for).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(cd "$HERE/../.." && pwd)"
KHRONOS_ROOT="${KHRONOS_ROOT:-/root/reference}"
OUT="$REPO/oracle/_ref"
SRC="$KHRONOS_ROOT/khronos/src"
for f in active_window/integration/tracking_integrator.cpp active_window/motion_detection/free_space_motion_detector.cpp utils/geometry_utils.cpp \
         active_window/object_detection/connected_semantics.cpp active_window/object_detection/instance_forwarding.cpp \
         active_window/tracking/max_iou_tracker.cpp active_window/data/track.cpp \
         active_window/tracking/external_tracker.cpp active_window/data/frame_data_buffer.cpp \
         backend/change_detection/ray_verificator.cpp backend/change_detection/ray_change_detector.cpp backend/change_state.cpp \
         backend/change_detection/background/ray_background_change_detector.cpp backend/change_detection/objects/ray_object_change_detector.cpp \
         active_window/object_extraction/mesh_object_extractor.cpp active_window/integration/object_integrator.cpp \
         active_window/active_window.cpp active_window/object_extraction/object_worker_pool.cpp; do
  [ -f "$SRC/$f" ] || { echo "build_ref.sh: $SRC/$f not found (no reference checkout here): keeping what is in $OUT" >&2; exit 3; }
done
mkdir -p "$OUT"
# (-fno-access-control: ref_harness.cpp's integrator bridge reads two private members of khronos::ObjectIntegrator;
#  liboracle.so: the CPU oracle stands behind hydra::ProjectiveIntegrator / MeshIntegrator, which /root/reference does not contain)
make -C "$REPO/oracle" -s
"${CXX:-g++}" -O2 -std=c++17 -ffp-contract=off -fPIC -shared -pthread -fno-access-control \
  -I"$HERE/standin" -I"$KHRONOS_ROOT/khronos/include" \
  "$HERE/ref_harness.cpp" \
  "$SRC/active_window/integration/tracking_integrator.cpp" \
  "$SRC/active_window/motion_detection/free_space_motion_detector.cpp" \
  "$SRC/utils/geometry_utils.cpp" \
  "$SRC/active_window/object_detection/connected_semantics.cpp" \
  "$SRC/active_window/object_detection/instance_forwarding.cpp" \
  "$SRC/active_window/tracking/max_iou_tracker.cpp" \
  "$SRC/active_window/data/track.cpp" \
  "$SRC/active_window/tracking/external_tracker.cpp" \
  "$SRC/active_window/data/frame_data_buffer.cpp" \
  "$SRC/backend/change_detection/ray_verificator.cpp" \
  "$SRC/backend/change_detection/ray_change_detector.cpp" \
  "$SRC/backend/change_state.cpp" \
  "$SRC/backend/change_detection/background/ray_background_change_detector.cpp" \
  "$SRC/backend/change_detection/objects/ray_object_change_detector.cpp" \
  "$SRC/active_window/object_extraction/mesh_object_extractor.cpp" \
  "$SRC/active_window/integration/object_integrator.cpp" \
  "$SRC/active_window/active_window.cpp" \
  "$SRC/active_window/object_extraction/object_worker_pool.cpp" \
  -L"$REPO/oracle" -loracle -Wl,-rpath,'$ORIGIN/..' \
  -o "$OUT/libref_khronos.so"
echo "built $OUT/libref_khronos.so"
