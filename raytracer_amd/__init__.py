"""MI355X-native trace loop for LastingLuck/Raytracer scenes (HIP kernels behind a C-ABI).

`raytracer_amd.api` binds libraytracer_amd.so; nothing here renders on the CPU.
"""
from .api import (  # noqa: F401
    RT_ACCEL_AUTO, RT_ACCEL_BRUTE, RT_ACCEL_LBVH, RT_ACCEL_LBVH_PACKET, RT_ACCEL_LBVH_WIDE, RT_ACCEL_LBVH_TEAM, RT_ACCEL_ALT_TREE, RT_ACCEL_SAH_TREE, RT_PROJ_ORTHO, RT_PROJ_PERSP, RT_TILE,
    RtError, RtStats, RtTreeStats, SceneArrays, Tracer, jitter_hash, load_library, output_name,
    parse_scene_file, render_multi, shard_bytes, shard_tiles, validate_scene, write_bmp24,
)
