"""ctypes mirror of include/hikari_mi355x.h (field order and types must match the header exactly)."""
import ctypes as C

c_f = C.c_float
c_i = C.c_int32
c_u = C.c_uint32
PF = C.POINTER(C.c_float)

HK_OK = 0
HK_ERR_INVALID, HK_ERR_DEVICE, HK_ERR_UNSUPPORTED = -1, -2, -3
HK_UNSET = -100       # hk_ctx_get_option: the knob has no value (not an error)

(HK_MAT_MATTE, HK_MAT_MIRROR, HK_MAT_GLASS, HK_MAT_CONDUCTOR, HK_MAT_COATED_DIFFUSE, HK_MAT_THIN_DIELECTRIC,
 HK_MAT_DIFFUSE_TRANSMISSION, HK_MAT_COATED_DIFFUSE_TRANSMISSION, HK_MAT_COATED_CONDUCTOR, HK_MAT_MIX,
 HK_MAT_FALLBACK) = range(11)
HK_MATF_REMAP_ROUGHNESS, HK_MATF_USE_ETA_K = 1, 2
(HK_LIGHT_POINT, HK_LIGHT_SPOT, HK_LIGHT_DIRECTIONAL, HK_LIGHT_SUN, HK_LIGHT_AMBIENT, HK_LIGHT_ENVIRONMENT,
 HK_LIGHT_DIFFUSE_AREA) = range(7)
HK_SPEC_RGB, HK_SPEC_ILLUMINANT = 0, 1
HK_TONEMAP_NONE, HK_TONEMAP_REINHARD, HK_TONEMAP_REINHARD_EXT, HK_TONEMAP_ACES, HK_TONEMAP_UNCHARTED2, HK_TONEMAP_FILMIC = range(6)
HK_MEDIUM_HOMOGENEOUS, HK_MEDIUM_GRID, HK_MEDIUM_RGB_GRID, HK_MEDIUM_NANOVDB = range(4)
HK_FILTER_BOX, HK_FILTER_TRIANGLE, HK_FILTER_GAUSSIAN, HK_FILTER_MITCHELL, HK_FILTER_LANCZOS = 1, 2, 3, 4, 5
HK_COHERENCE_NONE, HK_COHERENCE_SORTED, HK_COHERENCE_PER_TYPE = 0, 1, 2


class hk_texture(C.Structure):
    _fields_ = [("width", c_i), ("height", c_i), ("channels", c_i), ("kind", c_i), ("data", PF)]


class hk_tex_rgba(C.Structure):
    _fields_ = [("c", c_f * 4), ("tex", c_i)]


class hk_tex_f32(C.Structure):
    _fields_ = [("v", c_f), ("tex", c_i)]


class hk_material(C.Structure):
    _fields_ = [("kind", c_i), ("flags", c_i), ("rgb", hk_tex_rgba * 4), ("f", hk_tex_f32 * 8), ("i", c_i * 4),
                ("spectrum", c_i * 2), ("mix_key", c_u * 4)]


class hk_pl_spectrum(C.Structure):
    _fields_ = [("n", c_i), ("_pad", c_i), ("lambdas", PF), ("values", PF)]


class hk_medium_interface(C.Structure):
    _fields_ = [("material", c_i), ("inside", c_i), ("outside", c_i)]


class hk_tri_meta(C.Structure):
    _fields_ = [("medium_interface_idx", c_u), ("primitive_index", c_u), ("arealight_flat_idx_1based", c_u)]


class hk_light(C.Structure):
    _fields_ = [("kind", c_i), ("spectrum_kind", c_i), ("i_rgb", c_f * 4), ("poly", c_f * 3), ("illum_scale", c_f),
                ("scale", c_f), ("position", c_f * 3), ("direction", c_f * 3), ("world_to_light", c_f * 16),
                ("light_to_world", c_f * 16), ("cos_total_width", c_f), ("cos_falloff_start", c_f), ("v", c_f * 9),
                ("normal", c_f * 3), ("area", c_f), ("uv", c_f * 6), ("Le", hk_tex_rgba), ("two_sided", c_i),
                ("envmap", c_i)]


class hk_envmap(C.Structure):
    _fields_ = [("width", c_i), ("height", c_i), ("data", PF), ("rotation", c_f * 9), ("nu", c_i), ("nv", c_i),
                ("conditional_func", PF), ("conditional_cdf", PF), ("conditional_func_int", PF), ("marginal_func", PF),
                ("marginal_cdf", PF), ("marginal_func_int", c_f), ("_pad", c_i)]


class hk_medium(C.Structure):
    _fields_ = [("kind", c_i), ("sigma_a", c_f * 4), ("sigma_s", c_f * 4), ("Le", c_f * 4), ("g", c_f),
                ("sigma_scale", c_f), ("Le_scale", c_f), ("bounds_min", c_f * 3), ("bounds_max", c_f * 3),
                ("render_to_medium", c_f * 16), ("medium_to_render", c_f * 16), ("res", c_i * 3), ("density", PF),
                ("sigma_a_grid", PF), ("sigma_s_grid", PF), ("Le_grid", PF), ("majorant_res", c_i * 3),
                ("majorant", PF), ("max_density", c_f), ("nvdb_bytes", C.POINTER(C.c_uint8)), ("nvdb_size", C.c_int64),
                ("root_offset_1based", C.c_int64), ("upper_offset_1based", C.c_int64),
                ("lower_offset_1based", C.c_int64), ("leaf_offset_1based", C.c_int64), ("upper_count", c_i),
                ("lower_count", c_i), ("leaf_count", c_i), ("root_table_size", c_i), ("inv_mat", c_f * 9),
                ("vec", c_f * 3), ("index_bbox_min", c_i * 3), ("index_bbox_max", c_i * 3)]


class hk_scene_desc(C.Structure):
    _fields_ = [("n_triangles", c_i), ("n_materials", c_i), ("n_textures", c_i), ("n_media_interfaces", c_i),
                ("n_lights", c_i), ("n_envmaps", c_i), ("n_media", c_i), ("n_spectra", c_i), ("positions", PF),
                ("normals", PF), ("uvs", PF), ("tangents", PF), ("meta", C.POINTER(hk_tri_meta)),
                ("materials", C.POINTER(hk_material)), ("textures", C.POINTER(hk_texture)),
                ("media_interfaces", C.POINTER(hk_medium_interface)), ("lights", C.POINTER(hk_light)),
                ("envmaps", C.POINTER(hk_envmap)), ("media", C.POINTER(hk_medium)),
                ("spectra", C.POINTER(hk_pl_spectrum))]


class hk_tables(C.Structure):
    _fields_ = [("sobol_matrices", C.POINTER(c_u)), ("sobol_count", c_i), ("rgb2spec_res", c_i), ("cie_x", PF),
                ("cie_y", PF), ("cie_z", PF), ("rgb2spec_scale", PF), ("rgb2spec_coeffs", PF)]


class hk_integrator_params(C.Structure):
    _fields_ = [("max_depth", c_i), ("samples_per_pixel", c_i), ("russian_roulette_depth", c_i), ("regularize", c_i),
                ("material_coherence", c_i), ("max_component_value", c_f), ("filter_type", c_i),
                ("filter_radius", c_f * 2), ("filter_param1", c_f), ("filter_param2", c_f), ("accumulate_f64", c_i),
                ("sampler_seed", c_u), ("samples_per_pass", c_i)]


class hk_postprocess_params(C.Structure):
    _fields_ = [("exposure", c_f), ("tonemap", c_i), ("inv_gamma", c_f), ("apply_gamma", c_i), ("white_point", c_f),
                ("imaging_ratio", c_f), ("apply_wb", c_i), ("wb", c_f * 9), ("mask_escaped", c_i), ("bg", c_f * 3)]


class hk_denoise_params(C.Structure):
    _fields_ = [("iterations", c_i), ("sigma_color", c_f), ("sigma_normal", c_f), ("sigma_depth", c_f), ("use_variance", c_i)]


class hk_camera(C.Structure):
    _fields_ = [("raster_to_camera", c_f * 16), ("camera_to_world", c_f * 16), ("lens_radius", c_f),
                ("focal_distance", c_f), ("shutter_open", c_f), ("shutter_close", c_f), ("dx_camera", c_f * 3),
                ("dy_camera", c_f * 3)]


class hk_stats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("bvh_nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("hits_accepted", C.c_uint64), ("path_vertices", C.c_uint64),
                ("medium_collisions", C.c_uint64), ("light_bvh_nodes", C.c_uint64), ("seconds_trace", C.c_double),
                ("seconds_total", C.c_double), ("trace_launches", C.c_uint64), ("trace_nodes", C.c_uint64),
                ("trace_tris", C.c_uint64), ("shadow_nodes", C.c_uint64), ("shadow_tris", C.c_uint64),
                ("shadow_launches", C.c_uint64), ("shade_launches", C.c_uint64), ("seconds_shadow", C.c_double),
                ("seconds_shade", C.c_double), ("seconds_other", C.c_double), ("bytes_algorithmic_trace", C.c_uint64),
                ("bytes_algorithmic_shadow", C.c_uint64), ("bytes_algorithmic_shade", C.c_uint64), ("seconds_media", C.c_double),
                ("track_collisions", C.c_uint64), ("shadow_collisions", C.c_uint64), ("track_dda_steps", C.c_uint64),
                ("shadow_dda_steps", C.c_uint64), ("scatter_vertices", C.c_uint64), ("media_launches", C.c_uint64),
                ("bytes_algorithmic_media", C.c_uint64), ("seconds_select", C.c_double), ("select_launches", C.c_uint64),
                ("fused_passes", C.c_uint64)]


# every symbol include/hikari_mi355x.h declares (tests check the built library exports all of them)
EXPORTED_SYMBOLS = [
    "hk_ctx_create", "hk_ctx_destroy", "hk_last_error", "hk_ctx_set_tables", "hk_scene_create", "hk_scene_destroy",
    "hk_integrator_create", "hk_integrator_destroy", "hk_film_create", "hk_film_destroy", "hk_film_clear", "hk_render",
    "hk_film_read_rgb", "hk_film_read_accum", "hk_film_accum_device_ptr", "hk_sync", "hk_stats_get", "hk_stats_reset",
    "hk_stats_enable_counters", "hk_trace_closest", "hk_test_sobol", "hk_test_camera", "hk_test_uplift",
    "hk_test_light_bvh", "hk_test_bsdf", "hk_film_postprocess", "hk_film_fill_aux", "hk_postprocess", "hk_test_light", "hk_scene_bvh_info", "hk_scene_light_bvh_copy",
    "hk_denoise", "hk_test_mix", "hk_test_medium", "hk_test_trace_lean",
    "hk_render_tile", "hk_comm_create", "hk_comm_unique_id", "hk_comm_create_rank", "hk_comm_destroy", "hk_film_reduce",
    "hk_ctx_set_option", "hk_ctx_get_option", "hk_trim_cache", "hk_flush", "hk_film_read_rgb_async", "hk_film_read_wait",
    "hk_film_pin_host", "hk_film_unpin_host",
]
