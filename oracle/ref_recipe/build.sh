#!/bin/bash
# oracle/ref_recipe/build.sh -- builds and runs the reference vector generator (dump_vectors.cpp) against REAL upstream checkouts
# and leaves oracle/_ref/ref_small.npz, which tests/test_golden.py prefers over the oracle-generated tests/golden/aw_small.npz.
#
#   KHRONOS_ROOT           checkout of MIT-SPARK/Khronos            (default /root/reference)
#   HYDRA_ROOT             checkout of MIT-SPARK/Hydra @ main       (install/https.rosinstall:5-8)
#   SPATIAL_HASH_ROOT      checkout of MIT-SPARK/Spatial-Hash       (install/https.rosinstall:33-36)
#   CONFIG_UTILITIES_ROOT  checkout of MIT-SPARK/config_utilities   (install/https.rosinstall:1-4)
#   SPARK_DSG_ROOT         checkout of MIT-SPARK/Spark-DSG          (install/https.rosinstall:29-32)
#   DEP_FLAGS              compiler / linker flags of their own dependencies (Eigen, OpenCV, glog, yaml-cpp, ...), e.g.
#                          "$(pkg-config --cflags --libs eigen3 opencv4 libglog yaml-cpp)"
#
#   oracle/ref_recipe/build.sh --check     syntax check of the harness against the stand-in API headers (no checkouts needed; this
#                                          is what tests/test_cpu_oracle.py runs in the offline container)
#
# Nothing is copied from the checkouts into this repository; all outputs go to oracle/_ref/ (git-ignored).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REPO="$(cd "$HERE/../.." && pwd)"
OUT="$REPO/oracle/_ref"
CXX="${CXX:-g++}"
if [ "${1:-}" = "--check" ]; then
  exec "$CXX" -std=c++17 -fsyntax-only -Wall -Wextra -Wno-unused-parameter -I"$HERE/stub" "$HERE/dump_vectors.cpp"
fi
: "${HYDRA_ROOT:?set HYDRA_ROOT to a checkout of MIT-SPARK/Hydra (install/https.rosinstall:5-8)}"
: "${SPATIAL_HASH_ROOT:?set SPATIAL_HASH_ROOT to a checkout of MIT-SPARK/Spatial-Hash}"
: "${CONFIG_UTILITIES_ROOT:?set CONFIG_UTILITIES_ROOT to a checkout of MIT-SPARK/config_utilities}"
: "${SPARK_DSG_ROOT:?set SPARK_DSG_ROOT to a checkout of MIT-SPARK/Spark-DSG}"
KHRONOS_ROOT="${KHRONOS_ROOT:-/root/reference}"
mkdir -p "$OUT/npy"
INC="-I$KHRONOS_ROOT/khronos/include -I$HYDRA_ROOT/include -I$SPATIAL_HASH_ROOT/include -I$CONFIG_UTILITIES_ROOT/config_utilities/include -I$SPARK_DSG_ROOT/include"
# the translation units on the path (everything else of Hydra / Khronos stays out): the three Khronos files the harness calls
# into, and Hydra's reconstruction + input conversion sources.  A maintainer with a colcon workspace links the built libraries
# instead: set LINK_LIBS="-lhydra -lspatial_hash -lconfig_utilities -lspark_dsg" and leave SRC_HYDRA empty.
SRC_KHRONOS="$KHRONOS_ROOT/khronos/src/active_window/integration/tracking_integrator.cpp \
             $KHRONOS_ROOT/khronos/src/active_window/motion_detection/free_space_motion_detector.cpp \
             $KHRONOS_ROOT/khronos/src/utils/geometry_utils.cpp"
SRC_HYDRA="${SRC_HYDRA-$(ls "$HYDRA_ROOT"/src/reconstruction/*.cpp "$HYDRA_ROOT"/src/input/*.cpp 2>/dev/null | tr '\n' ' ')}"
"$CXX" -O2 -std=c++17 -ffp-contract=off -pthread $INC ${DEP_FLAGS:-} -o "$OUT/dump_vectors" \
  "$HERE/dump_vectors.cpp" "$REPO/khronos_amd/synth/synth.cpp" $SRC_KHRONOS $SRC_HYDRA ${LINK_LIBS:-}
"$OUT/dump_vectors" "$OUT/npy"
python3 - "$OUT" <<'PY'
import glob, os, sys
import numpy as np
out = sys.argv[1]
arrays = {os.path.splitext(os.path.basename(f))[0]: np.load(f) for f in glob.glob(os.path.join(out, "npy", "*.npy"))}
arrays.update(W=96, H=72, N=16, provenance=np.array("upstream: hydra::ProjectiveIntegrator / MeshIntegrator + khronos::TrackingIntegrator / FreeSpaceMotionDetector (oracle/ref_recipe/dump_vectors.cpp)"))
np.savez_compressed(os.path.join(out, "ref_small.npz"), **arrays)
print("wrote", os.path.join(out, "ref_small.npz"), sorted(arrays))
PY
# which setting of the ASSUMPTIONS.md [A] switches reproduces these vectors (written next to them: ref_small.npz.switches.json)
python3 "$HERE/match_switches.py" "$OUT/ref_small.npz" --eps 1e-3
