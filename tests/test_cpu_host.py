"""-m "not gpu": host-side logic of the khronos::ActiveWindow mirror (no device needed)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELFTEST = os.path.join(ROOT, "khronos_amd", "lib", "host_selftest")
REF_CFG = "/root/reference/khronos_ros/config/mapper/uHumans2.yaml"

UHUMANS2_ACTIVE_WINDOW = """
# keys and values of khronos_ros/config/mapper/uHumans2.yaml:3-100 (shared anchors + active_window block)
shared_parameters:
  max_range: &max_range 5 # m
  temporal_window: &temporal_window 3 # s
  active_window_threads: &active_window_threads -1
active_window:
  type: "ActiveWindow"
  verbosity: 2
  min_output_separation: 0.4 # s
  frame_data_buffer:
    max_buffer_size: 300
    store_every_n_frames: 1
  volumetric_map:
    voxel_size: 0.1 # m
    truncation_distance: 0.2 # m (Usually 2-3x voxel size)
    voxels_per_side: 16
    with_semantics: true  # Enable semantic layer for object detection
  motion_detector:
    type: "FreeSpaceMotionDetector" # 'FreeSpaceMotionDetector'
    min_cluster_size: 500 # pixels
    min_separation_distance: 2 # voxels
    num_threads: *active_window_threads
    max_range: *max_range # m
  object_detector:
    type: "ConnectedSemantics"
    min_cluster_size: 50 # pixels
  tracker:
    verbosity: 0
    type: "MaxIouTracker"
  projective_integrator:
    num_threads: *active_window_threads
  tracking_integrator:
    temporal_window: *temporal_window
    num_threads: *active_window_threads
  object_extractor:
    type: MeshObjectExtractor
    min_object_volume: 0.005 # m^3
    max_object_volume: 10.0 # m^3
    only_extract_reconstructed_objects: true # used to be false
    object_reconstruction_resolution: -0.02
    projective_integrator:
      num_threads: 8
"""

EXPECT = {"voxel_size": 0.1, "truncation_distance": 0.2, "voxels_per_side": 16, "with_semantics": 1,
          "min_output_separation": 0.4, "motion_detector": "FreeSpaceMotionDetector", "md_min_cluster_size": 500,
          "md_min_separation_distance": 2, "md_max_range": 5, "temporal_window": 3, "object_extractor": "MeshObjectExtractor",
          "min_object_volume": 0.005, "max_buffer_size": 300, "object_detector": "ConnectedSemantics", "tracker": "MaxIouTracker",
          "only_extract_reconstructed_objects": 1, "object_reconstruction_resolution": -0.02}


def _parse(path):
    out = subprocess.run([SELFTEST, str(path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout)


def test_host_selftest_known_answers():
    out = subprocess.run([SELFTEST], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "host selftest ok" in out.stdout


def test_yaml_loader_reads_reference_keys(tmp_path):
    p = tmp_path / "aw.yaml"
    p.write_text(UHUMANS2_ACTIVE_WINDOW)
    got = _parse(p)
    for k, v in EXPECT.items():
        assert got[k] == (v if isinstance(v, str) else __import__("pytest").approx(v, rel=1e-6)), k


def test_yaml_loader_on_reference_file_when_present():
    # /root/reference only exists in the dev container; the committed copy of its keys above covers the GPU box
    if not os.path.exists(REF_CFG):
        return
    got = _parse(REF_CFG)
    for k, v in EXPECT.items():
        assert got[k] == (v if isinstance(v, str) else __import__("pytest").approx(v, rel=1e-6)), k
