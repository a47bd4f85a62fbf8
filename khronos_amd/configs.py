"""Configurations shared by bench.py and the parity tests (no torch import here: the tests load the HIP library first)."""

# `active_window:` YAML of the object half (keys and values of khronos_ros/config/mapper/uHumans2.yaml:35-100; the
# object labels are the scene's primitives, the mover included)
OBJECT_YAML = """
active_window:
  type: "ActiveWindow"
  min_output_separation: 0.4
  frame_data_buffer:
    max_buffer_size: %(buf)d
    store_every_n_frames: 1
  volumetric_map:
    voxel_size: %(vs)r
    truncation_distance: %(trunc)r
    voxels_per_side: 16
    with_semantics: true
  object_detector:
    type: "ConnectedSemantics"
    min_cluster_size: 50
    use_full_connectivity: true
    use_3d: true
    grid_size: 0.1
    max_range: 5
    object_labels: [7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19]
  tracker:
    type: "MaxIouTracker"
    track_by: "voxels"
    min_semantic_iou: 0.25
    min_cross_iou: 0.1
    voxel_size: 0.2
    temporal_window: 3
    min_num_observations: 15
  object_extractor:
    type: MeshObjectExtractor
    min_object_allocation_confidence: 0.5
    min_object_volume: 0.005
    max_object_volume: 10.0
    min_dynamic_displacement: 1
    only_extract_reconstructed_objects: true
    min_object_reconstruction_confidence: 0.5
    min_object_reconstruction_observations: 0
    object_reconstruction_resolution: -0.02
"""

# the whole `active_window:` mapping for khronos::ActiveWindow (libkhronos_amd_host.so) at a bench preset: OBJECT_YAML plus the
# motion detector / integrators (khronos_ros/config/mapper/uHumans2.yaml:40-57) and the device sizing block -- what
# `aw_demo --bench` (bench.py stream cxx_active_window) runs
ACTIVE_WINDOW_YAML = OBJECT_YAML + """  motion_detector:
    type: "FreeSpaceMotionDetector"
    min_cluster_size: 500
    min_separation_distance: 2
    max_range: 5
  projective_integrator:
    num_threads: -1
  tracking_integrator:
    num_threads: -1
  device:
    num_labels: %(labels)d
    max_blocks: %(max_blocks)d
"""
