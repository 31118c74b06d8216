"""MI355X-native path-tracing hot path — Python host bindings over the C ABI (include/rt_hip.h).

The directory name has a hyphen, so import it through ``rta.load()`` at the repo root (it registers
this package as ``ray_tracer_archive_amd``).  Importing the package never touches the GPU; creating
a ``Context`` does, and fails loudly when the HIP library or a device is missing — there is no CPU
fallback in the product path.
"""
from . import _abi  # noqa: F401
from .api import (Context, Scene, HostScene, RtError, camera_new, lib, lib_path, tonemap, write_color, write_png, write_image, untile,
                  output_floats, make_params, compile_info, compile_dump, MultiContext, MultiScene, comm_unique_id, untile_rgb8, wide_layout_check, runtime_libraries, upload_options)  # noqa: F401
from .scene import SceneBuilder  # noqa: F401
from .scene_json import JsonScene, load_scene  # noqa: F401
