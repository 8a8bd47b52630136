# HikariMI355X.jl — the reference-side binding a Hikari.jl maintainer adds (INTEGRATION.md): a `Hikari.Integrator` subtype that
# flattens `Scene` / `Film` / `Camera` into the POD records of include/hikari_mi355x.h and `ccall`s libhikari_mi355x.so.
#
# STATUS: written against the reference sources under /root/reference (file:line cited at each step) and against the C header;
# it has NOT been executed — the build image has no Julia toolchain and Raycore.jl (the un-vendored BVH dependency, branch
# `sd/multitype-vec`, Project.toml:33-34) is absent.  What is checked here instead: tests/test_julia_shim.py parses the `struct Hk*`
# declarations below and compares every field offset and size with the C header (through the ctypes mirror, itself checked
# against `sizeof` from gcc), and checks that every `ccall` names an exported symbol with the right argument count.
# Raycore internals this file touches, all of them through names Hikari's own sources use:
#   accel.instances[i].transform / .inv_transform        src/surface_interaction.jl:418-423
#   accel.blas_array                                      src/scene.jl:196-197
#   blas.primitives (Raycore.BVH)                         src/scene.jl:201
#   Raycore.vertices / normals / uvs / tangents, triangle.metadata::TriangleMeta   src/surface_interaction.jl:331-353, src/scene.jl:11-15
#   Raycore.get_static(multitypeset), Raycore.deref(static, ::TextureRef), SetKey.type_idx / .vec_idx, Raycore.is_valid
# The ONE unknown is the name of the field that links a TLAS instance to its BLAS: `instance_blas` resolves it at run time from a
# short list and fails with an explicit message otherwise.
module HikariMI355X

using Hikari
import Hikari: Integrator, render!, clear!
using Raycore
using LinearAlgebra: normalize, transpose

const LIB = get(ENV, "HIKARI_MI355X_LIB", "libhikari_mi355x")
const Vec3f = Hikari.Vec3f

check(st::Int32, what) = st == 0 || error("$what failed ($st): " * unsafe_string(ccall((:hk_last_error, LIB), Cstring, ())))

# ---- POD mirrors of include/hikari_mi355x.h (field order and types must match; tests/test_julia_shim.py checks the layout) ----
struct HkTexture
    width::Int32; height::Int32; channels::Int32; kind::Int32
    data::Ptr{Float32}
end
struct HkTexRgba
    c::NTuple{4,Float32}; tex::Int32
end
struct HkTexF32
    v::Float32; tex::Int32
end
struct HkMaterial
    kind::Int32; flags::Int32
    rgb::NTuple{4,HkTexRgba}
    f::NTuple{8,HkTexF32}
    i::NTuple{4,Int32}
    spectrum::NTuple{2,Int32}
    mix_key::NTuple{4,UInt32}
end
struct HkPlSpectrum
    n::Int32; _pad::Int32
    lambdas::Ptr{Float32}; values::Ptr{Float32}
end
struct HkMediumInterface
    material::Int32; inside::Int32; outside::Int32
end
struct HkTriMeta
    medium_interface_idx::UInt32; primitive_index::UInt32; arealight_flat_idx_1based::UInt32
end
struct HkLight
    kind::Int32; spectrum_kind::Int32
    i_rgb::NTuple{4,Float32}
    poly::NTuple{3,Float32}
    illum_scale::Float32
    scale::Float32
    position::NTuple{3,Float32}
    direction::NTuple{3,Float32}
    world_to_light::NTuple{16,Float32}
    light_to_world::NTuple{16,Float32}
    cos_total_width::Float32; cos_falloff_start::Float32
    v::NTuple{9,Float32}
    normal::NTuple{3,Float32}
    area::Float32
    uv::NTuple{6,Float32}
    Le::HkTexRgba
    two_sided::Int32
    envmap::Int32
end
struct HkEnvmap
    width::Int32; height::Int32
    data::Ptr{Float32}
    rotation::NTuple{9,Float32}
    nu::Int32; nv::Int32
    conditional_func::Ptr{Float32}; conditional_cdf::Ptr{Float32}; conditional_func_int::Ptr{Float32}
    marginal_func::Ptr{Float32}; marginal_cdf::Ptr{Float32}
    marginal_func_int::Float32
    _pad::Int32
end
struct HkMedium
    kind::Int32
    sigma_a::NTuple{4,Float32}; sigma_s::NTuple{4,Float32}; Le::NTuple{4,Float32}
    g::Float32
    sigma_scale::Float32; Le_scale::Float32
    bounds_min::NTuple{3,Float32}; bounds_max::NTuple{3,Float32}
    render_to_medium::NTuple{16,Float32}; medium_to_render::NTuple{16,Float32}
    res::NTuple{3,Int32}
    density::Ptr{Float32}
    sigma_a_grid::Ptr{Float32}; sigma_s_grid::Ptr{Float32}; Le_grid::Ptr{Float32}
    majorant_res::NTuple{3,Int32}
    majorant::Ptr{Float32}
    max_density::Float32
    nvdb_bytes::Ptr{UInt8}
    nvdb_size::Int64
    root_offset_1based::Int64; upper_offset_1based::Int64; lower_offset_1based::Int64; leaf_offset_1based::Int64
    upper_count::Int32; lower_count::Int32; leaf_count::Int32; root_table_size::Int32
    inv_mat::NTuple{9,Float32}; vec::NTuple{3,Float32}
    index_bbox_min::NTuple{3,Int32}; index_bbox_max::NTuple{3,Int32}
end
struct HkSceneDesc
    n_triangles::Int32; n_materials::Int32; n_textures::Int32; n_media_interfaces::Int32
    n_lights::Int32; n_envmaps::Int32; n_media::Int32; n_spectra::Int32
    positions::Ptr{Float32}; normals::Ptr{Float32}; uvs::Ptr{Float32}; tangents::Ptr{Float32}
    meta::Ptr{HkTriMeta}
    materials::Ptr{HkMaterial}
    textures::Ptr{HkTexture}
    media_interfaces::Ptr{HkMediumInterface}
    lights::Ptr{HkLight}
    envmaps::Ptr{HkEnvmap}
    media::Ptr{HkMedium}
    spectra::Ptr{HkPlSpectrum}
end
struct HkTables
    sobol_matrices::Ptr{UInt32}; sobol_count::Int32; rgb2spec_res::Int32
    cie_x::Ptr{Float32}; cie_y::Ptr{Float32}; cie_z::Ptr{Float32}
    rgb2spec_scale::Ptr{Float32}; rgb2spec_coeffs::Ptr{Float32}
end
struct HkIntegratorParams
    max_depth::Int32; samples_per_pixel::Int32; russian_roulette_depth::Int32; regularize::Int32
    material_coherence::Int32; max_component_value::Float32; filter_type::Int32
    filter_radius::NTuple{2,Float32}; filter_param1::Float32; filter_param2::Float32
    accumulate_f64::Int32; sampler_seed::UInt32; samples_per_pass::Int32
end
struct HkCamera
    raster_to_camera::NTuple{16,Float32}; camera_to_world::NTuple{16,Float32}
    lens_radius::Float32; focal_distance::Float32; shutter_open::Float32; shutter_close::Float32
    dx_camera::NTuple{3,Float32}; dy_camera::NTuple{3,Float32}
end
struct HkPostprocessParams
    exposure::Float32; tonemap::Int32; inv_gamma::Float32; apply_gamma::Int32; white_point::Float32
    imaging_ratio::Float32; apply_wb::Int32; wb::NTuple{9,Float32}; mask_escaped::Int32; bg::NTuple{3,Float32}
end
struct HkDenoiseParams
    iterations::Int32; sigma_color::Float32; sigma_normal::Float32; sigma_depth::Float32; use_variance::Int32
end

# enum values of the header
const HK_MAT_MATTE, HK_MAT_MIRROR, HK_MAT_GLASS, HK_MAT_CONDUCTOR, HK_MAT_COATED_DIFFUSE, HK_MAT_THIN_DIELECTRIC = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4), Int32(5)
const HK_MAT_DIFFUSE_TRANSMISSION, HK_MAT_COATED_DIFFUSE_TRANSMISSION, HK_MAT_COATED_CONDUCTOR, HK_MAT_MIX, HK_MAT_FALLBACK = Int32(6), Int32(7), Int32(8), Int32(9), Int32(10)
const HK_MATF_REMAP_ROUGHNESS, HK_MATF_USE_ETA_K = Int32(1), Int32(2)
const HK_LIGHT_POINT, HK_LIGHT_SPOT, HK_LIGHT_DIRECTIONAL, HK_LIGHT_SUN, HK_LIGHT_AMBIENT, HK_LIGHT_ENVIRONMENT, HK_LIGHT_DIFFUSE_AREA = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4), Int32(5), Int32(6)
const HK_SPEC_RGB, HK_SPEC_ILLUMINANT = Int32(0), Int32(1)
const HK_MEDIUM_HOMOGENEOUS, HK_MEDIUM_GRID, HK_MEDIUM_RGB_GRID, HK_MEDIUM_NANOVDB = Int32(0), Int32(1), Int32(2), Int32(3)

# ---------------------------------------------------------------------------------------------------------------------------
mutable struct DeviceState
    device::Int
    ctx::Ptr{Cvoid}; integ::Ptr{Cvoid}; film::Ptr{Cvoid}; scene::Ptr{Cvoid}
end

mutable struct MI355XVolPath <: Integrator
    params::HkIntegratorParams
    samples_per_pixel::Int32
    devs::Vector{DeviceState}
    comm::Ptr{Cvoid}
    scene_id::UInt
    film_size::Tuple{Int,Int}
    display::Symbol            # what `render!` does with film.framebuffer after its sample: :every, :pipelined, :manual (see MI355XVolPath)
    display_every::Int         # ... and every how many calls
    calls_since_display::Int
    read_pending::Bool         # an hk_film_read_rgb_async nobody has waited for yet
    pinned_fb::Any             # the framebuffer registered with hk_film_pin_host (held here so that it outlives the registration), or nothing
end

rowmajor(m) = ntuple(i -> Float32(m[(i - 1) ÷ 4 + 1, (i - 1) % 4 + 1]), 16)
rowmajor3(m) = ntuple(i -> Float32(m[(i - 1) ÷ 3 + 1, (i - 1) % 3 + 1]), 9)
tup3(v) = (Float32(v[1]), Float32(v[2]), Float32(v[3]))
rgba(s::Hikari.RGBSpectrum) = (s.c[1], s.c[2], s.c[3], s.c[4])

"""
    MI355XVolPath(; max_depth=8, samples=64, russian_roulette_depth=3, regularize=true, material_coherence=:none,
                  max_component_value=10f0, filter=GaussianFilter(), accumulation_eltype=Float32, devices=0:0)

Same keywords and defaults as `Hikari.VolPath` (volpath.jl:75-101) plus `devices`: the GPUs of this node that share a frame
(sample-index sharding + one RCCL reduce of the film inside the library, SURVEY 8e), and how `render!` — ONE sample per call, what
RayMakie's interactive loop drives (volpath.jl:445-450) — keeps `film.framebuffer` (host memory here; a device array in the reference):

  * `display = :every` (default), `display_every = k`: the frame is read back after every k-th call (`hk_film_read_rgb`: K13 + a copy into
    `film.framebuffer`, which the shim registers with `hk_film_pin_host` and keeps alive).  With k = 1 `film.framebuffer` is current after every call,
    exactly like the reference; calls that are not followed by a read-back are only NOTED by the library and rendered as one pass with the
    next read-back (bit-identical film, up to 7x less time per sample) — k = 4 … 16 is what a viewer at 60 Hz wants.
  * `display = :pipelined`: after call i `film.framebuffer` holds the frame of call i − 1; the copy of frame i is in flight while call
    i + 1 renders (`hk_film_read_rgb_async` / `hk_film_read_wait`), so the GPU never waits for the host.  `sync_display!(vp, film)` brings
    the last frame in.
  * `display = :manual`: `render!` never reads back; call `sync_display!(vp, film)` when a frame is wanted (an offline loop).
The functor `vp(scene, film, camera)` always returns with the finished frame in place.
"""
function MI355XVolPath(; max_depth::Int = 8, samples::Int = 64, russian_roulette_depth::Int = 3, regularize::Bool = true,
                       material_coherence::Symbol = :none, max_component_value::Real = 10f0,
                       filter = Hikari.GaussianFilter(), accumulation_eltype::DataType = Float32,
                       devices = 0:0, display::Symbol = :every, display_every::Int = 1)
    @assert display in (:every, :pipelined, :manual) && display_every >= 1
    @assert material_coherence in (:none, :sorted, :per_type)          # volpath.jl:85
    @assert accumulation_eltype in (Float32, Float64)                 # volpath.jl:86
    fp = Hikari.GPUFilterParams(filter)                                # filter.jl:574-604
    p = HkIntegratorParams(max_depth, samples, russian_roulette_depth, regularize, findfirst(==(material_coherence), (:none, :sorted, :per_type)) - 1,
                           Float32(max_component_value), fp.filter_type, (fp.radius[1], fp.radius[2]), fp.param1, fp.param2,
                           accumulation_eltype === Float64, UInt32(0), 0)
    devs = [DeviceState(Int(d), C_NULL, C_NULL, C_NULL, C_NULL) for d in devices]
    MI355XVolPath(p, samples, devs, C_NULL, UInt(0), (0, 0), display, display_every, 0, false, nothing)
end

function ensure_ctx!(vp::MI355XVolPath)
    vp.devs[1].ctx != C_NULL && return
    tab = Hikari.get_srgb_table()
    sob, cx, cy, cz = Hikari.SobolMatrices32, Hikari.CIE_X, Hikari.CIE_Y, Hikari.CIE_Z
    scale, coeffs = Vector{Float32}(tab.scale), Array{Float32}(tab.coeffs)
    for d in vp.devs
        r = Ref{Ptr{Cvoid}}()
        check(ccall((:hk_ctx_create, LIB), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), d.device, C_NULL, r), "hk_ctx_create")
        d.ctx = r[]
        GC.@preserve sob cx cy cz scale coeffs begin
            t = HkTables(pointer(sob), length(sob), tab.res, pointer(cx), pointer(cy), pointer(cz), pointer(scale), pointer(coeffs))
            check(ccall((:hk_ctx_set_tables, LIB), Int32, (Ptr{Cvoid}, Ref{HkTables}), d.ctx, t), "hk_ctx_set_tables")
        end
        ri = Ref{Ptr{Cvoid}}()
        check(ccall((:hk_integrator_create, LIB), Int32, (Ptr{Cvoid}, Ref{HkIntegratorParams}, Ref{Ptr{Cvoid}}), d.ctx, vp.params, ri), "hk_integrator_create")
        d.integ = ri[]
    end
    if length(vp.devs) > 1
        ctxs = [d.ctx for d in vp.devs]
        rc = Ref{Ptr{Cvoid}}()
        GC.@preserve ctxs check(ccall((:hk_comm_create, LIB), Int32, (Ptr{Ptr{Cvoid}}, Int32, Ref{Ptr{Cvoid}}), pointer(ctxs), length(ctxs), rc), "hk_comm_create")
        vp.comm = rc[]
    end
    nothing
end

# ---------------------------------------------------------------------------------------------------------------------------
# Scene flattening: Hikari.Scene -> hk_scene_desc (the record-by-record twin of hikari.jl_amd/scene.py::Scene.sync)
# ---------------------------------------------------------------------------------------------------------------------------
"The BLAS (Raycore.BVH over the instance's triangles) of a TLAS instance; the linking field is resolved from the names Raycore has used."
function instance_blas(accel, inst)
    for f in (:blas_index, :blas_idx, :blas_id, :geometry_index, :geometry_id, :blas)
        hasproperty(inst, f) || continue
        v = getproperty(inst, f)
        return v isa Integer ? accel.blas_array[v] : v
    end
    error("HikariMI355X: cannot find the BLAS of a TLAS instance (fields: $(propertynames(inst))); adapt `instance_blas` to this Raycore version")
end

"Transformation applied to a point: divide by w unless w == 1 (Raycore `Transformation(p::Point3f)`)."
function xform_point(m, p)
    x = m[1, 1] * p[1] + m[1, 2] * p[2] + m[1, 3] * p[3] + m[1, 4]
    y = m[2, 1] * p[1] + m[2, 2] * p[2] + m[2, 3] * p[3] + m[2, 4]
    z = m[3, 1] * p[1] + m[3, 2] * p[2] + m[3, 3] * p[3] + m[3, 4]
    w = m[4, 1] * p[1] + m[4, 2] * p[2] + m[4, 3] * p[3] + m[4, 4]
    w == 1f0 ? (Float32(x), Float32(y), Float32(z)) : (Float32(x / w), Float32(y / w), Float32(z / w))
end
"Normals go through the inverse transpose and are re-normalised (surface_interaction.jl:425-436); NaN (= absent) stays NaN."
function xform_normal(inv_m, n)
    any(isnan, n) && return (NaN32, NaN32, NaN32)
    v = Vec3f(inv_m[1, 1] * n[1] + inv_m[2, 1] * n[2] + inv_m[3, 1] * n[3],
              inv_m[1, 2] * n[1] + inv_m[2, 2] * n[2] + inv_m[3, 2] * n[3],
              inv_m[1, 3] * n[1] + inv_m[2, 3] * n[2] + inv_m[3, 3] * n[3])
    tup3(normalize(v))
end
function xform_dir(m, d)
    any(isnan, d) && return (NaN32, NaN32, NaN32)
    tup3(normalize(Vec3f(m[1, 1] * d[1] + m[1, 2] * d[2] + m[1, 3] * d[3], m[2, 1] * d[1] + m[2, 2] * d[2] + m[2, 3] * d[3], m[3, 1] * d[1] + m[3, 2] * d[2] + m[3, 3] * d[3])))
end

"Builder state: arrays handed to the library are collected in `keep` so that one GC.@preserve covers the hk_scene_create call."
mutable struct Flattener
    static_mats                     # StaticMultiTypeSet the TextureRefs being resolved point into (materials, then lights)
    textures::Vector{HkTexture}
    tex_ids::Dict{UInt,Int32}
    spectra::Vector{HkPlSpectrum}
    envmaps::Vector{HkEnvmap}
    keep::Vector{Any}
end

function texture_index!(fl::Flattener, data::AbstractArray, kind::Int32 = Int32(0))
    get!(fl.tex_ids, objectid(data)) do
        T = eltype(data)
        arr, ch = if T === Float32
            (Array{Float32}(data), Int32(1))
        elseif T === Hikari.RGBSpectrum
            (Array{Hikari.RGBSpectrum}(data), Int32(4))                # 4 floats per texel (spectrum.jl:38-43)
        else                                                           # RGB{Float32} and friends: promote to RGBSpectrum
            (map(c -> Hikari.RGBSpectrum(Float32(c.r), Float32(c.g), Float32(c.b), 1f0), data), Int32(4))
        end
        push!(fl.keep, arr)
        h, w = size(arr, 1), size(arr, 2)                              # Julia [height, width] column-major == the ABI's texture layout
        push!(fl.textures, HkTexture(w, h, ch, kind, Ptr{Float32}(pointer(arr))))
        Int32(length(fl.textures) - 1)
    end
end

"A material parameter: constant or texture (texture-ref.jl:50-84).  Returns (constant, texture index or -1)."
function resolve_tex(fl::Flattener, x)
    x isa Raycore.TextureRef && (x = Raycore.deref(fl.static_mats, x))
    x isa Hikari.Texture && (x = x.isconst ? x.constval : x.data)
    if x isa Hikari.VertexColorTexture                               # textures/basic.jl:42-46, texture-ref.jl:230-235
        fc = x.face_colors
        fc isa Raycore.TextureRef && (fc = Raycore.deref(fl.static_mats, fc))
        return (nothing, texture_index!(fl, fc, Int32(1)))
    end
    x isa AbstractArray && ndims(x) == 0 && (x = x[])
    x isa AbstractArray && return (nothing, texture_index!(fl, x))
    return (x, Int32(-1))
end
function tex_rgba(fl::Flattener, x)
    c, t = resolve_tex(fl, x)
    c === nothing && return HkTexRgba((0f0, 0f0, 0f0, 1f0), t)
    c isa Hikari.RGBSpectrum && return HkTexRgba(rgba(c), Int32(-1))
    c isa Real && return HkTexRgba((Float32(c), Float32(c), Float32(c), 1f0), Int32(-1))
    return HkTexRgba((Float32(c.r), Float32(c.g), Float32(c.b), 1f0), Int32(-1))      # RGB{Float32}
end
function tex_f32(fl::Flattener, x)
    c, t = resolve_tex(fl, x)
    c === nothing ? HkTexF32(0f0, t) : HkTexF32(Float32(c), Int32(-1))
end
const NO_RGBA = HkTexRgba((0f0, 0f0, 0f0, 1f0), Int32(-1))
const NO_F32 = HkTexF32(0f0, Int32(-1))
pad4(xs...) = ntuple(i -> i <= length(xs) ? xs[i] : NO_RGBA, 4)
pad8(xs...) = ntuple(i -> i <= length(xs) ? xs[i] : NO_F32, 8)
f32c(v) = HkTexF32(Float32(v), Int32(-1))

function spectrum_index!(fl::Flattener, s::Hikari.PiecewiseLinearSpectrum)
    lam, val = collect(Float32, s.lambdas), collect(Float32, s.values)
    push!(fl.keep, lam, val)
    push!(fl.spectra, HkPlSpectrum(length(lam), 0, pointer(lam), pointer(val)))
    Int32(length(fl.spectra) - 1)
end
"eta / k of the conductors: measured spectrum or colour (uber-material.jl:378-384)"
function ior_slot(fl::Flattener, x)
    x isa Hikari.PiecewiseLinearSpectrum ? (NO_RGBA, spectrum_index!(fl, x)) : (tex_rgba(fl, x), Int32(-1))
end

"One hk_material per scene material (Appendix A of SURVEY.md).  `flat` maps a SetKey to the flat 0-based material index."
function material_record(fl::Flattener, m, flat)
    z4, nosp, nokey = (Int32(0), Int32(0), Int32(0), Int32(0)), (Int32(-1), Int32(-1)), (UInt32(0), UInt32(0), UInt32(0), UInt32(0))
    remap(x) = x.remap_roughness ? HK_MATF_REMAP_ROUGHNESS : Int32(0)
    if m isa Hikari.MatteMaterial
        return HkMaterial(HK_MAT_MATTE, 0, pad4(tex_rgba(fl, m.Kd)), pad8(tex_f32(fl, m.σ)), z4, nosp, nokey)
    elseif m isa Hikari.MirrorMaterial
        return HkMaterial(HK_MAT_MIRROR, 0, pad4(tex_rgba(fl, m.Kr)), pad8(), z4, nosp, nokey)
    elseif m isa Hikari.GlassMaterial
        return HkMaterial(HK_MAT_GLASS, remap(m), pad4(tex_rgba(fl, m.Kr), tex_rgba(fl, m.Kt)), pad8(tex_f32(fl, m.index)), z4, nosp, nokey)
    elseif m isa Hikari.ConductorMaterial
        (e, se), (k, sk) = ior_slot(fl, m.eta), ior_slot(fl, m.k)
        return HkMaterial(HK_MAT_CONDUCTOR, remap(m), pad4(e, k), pad8(tex_f32(fl, m.roughness)), z4, (se, sk), nokey)
    elseif m isa Hikari.CoatedDiffuseMaterial
        return HkMaterial(HK_MAT_COATED_DIFFUSE, remap(m), pad4(tex_rgba(fl, m.reflectance), tex_rgba(fl, m.albedo)),
                          pad8(tex_f32(fl, m.u_roughness), tex_f32(fl, m.v_roughness), tex_f32(fl, m.thickness), f32c(m.eta), tex_f32(fl, m.g)),
                          (m.max_depth, m.n_samples, Int32(0), Int32(0)), nosp, nokey)
    elseif m isa Hikari.ThinDielectricMaterial
        return HkMaterial(HK_MAT_THIN_DIELECTRIC, 0, pad4(), pad8(f32c(m.eta)), z4, nosp, nokey)
    elseif m isa Hikari.DiffuseTransmissionMaterial
        return HkMaterial(HK_MAT_DIFFUSE_TRANSMISSION, 0, pad4(tex_rgba(fl, m.reflectance), tex_rgba(fl, m.transmittance)), pad8(f32c(m.scale)), z4, nosp, nokey)
    elseif m isa Hikari.CoatedDiffuseTransmissionMaterial
        return HkMaterial(HK_MAT_COATED_DIFFUSE_TRANSMISSION, remap(m), pad4(tex_rgba(fl, m.reflectance), tex_rgba(fl, m.transmittance), tex_rgba(fl, m.albedo)),
                          pad8(tex_f32(fl, m.u_roughness), tex_f32(fl, m.v_roughness), tex_f32(fl, m.thickness), f32c(m.eta), tex_f32(fl, m.g)),
                          (m.max_depth, m.n_samples, Int32(0), Int32(0)), nosp, nokey)
    elseif m isa Hikari.CoatedConductorMaterial
        (e, se), (k, sk) = ior_slot(fl, m.conductor_eta), ior_slot(fl, m.conductor_k)
        flags = remap(m) | (m.use_eta_k ? HK_MATF_USE_ETA_K : Int32(0))
        return HkMaterial(HK_MAT_COATED_CONDUCTOR, flags, pad4(e, k, tex_rgba(fl, m.reflectance), tex_rgba(fl, m.albedo)),
                          pad8(tex_f32(fl, m.interface_u_roughness), tex_f32(fl, m.interface_v_roughness), f32c(m.interface_eta),
                               tex_f32(fl, m.conductor_u_roughness), tex_f32(fl, m.conductor_v_roughness), tex_f32(fl, m.thickness), tex_f32(fl, m.g)),
                          (m.max_depth, m.n_samples, Int32(0), Int32(0)), (se, sk), nokey)
    elseif m isa Hikari.MixMaterial                                    # mix-material.jl: children by SetKey, hashed with their keys (:96-127)
        k1, k2 = m.material1_idx, m.material2_idx
        return HkMaterial(HK_MAT_MIX, 0, pad4(), pad8(tex_f32(fl, m.amount)), (flat(k1), flat(k2), Int32(0), Int32(0)), nosp,
                          (UInt32(k1.type_idx), UInt32(k1.vec_idx), UInt32(k2.type_idx), UInt32(k2.vec_idx)))
    end
    # any other Material (a bare Emissive, ...): no BSDF method => the gray 0.5 Lambertian fallback (quirk Q24)
    return HkMaterial(HK_MAT_FALLBACK, 0, pad4(), pad8(), z4, nosp, nokey)
end

"Light spectrum: RGBSpectrum (uplifted at run time) or a baked RGBIlluminantSpectrum (rgb2spec.jl:317-335)"
function spectrum_fields(i)
    if i isa Hikari.RGBIlluminantSpectrum
        return (HK_SPEC_ILLUMINANT, (0f0, 0f0, 0f0, 1f0), (i.poly.c0, i.poly.c1, i.poly.c2), i.scale)
    end
    return (HK_SPEC_RGB, rgba(i), (0f0, 0f0, 0f0), 0f0)
end
const Z3, Z6, Z9, Z16 = (0f0, 0f0, 0f0), ntuple(_ -> 0f0, 6), ntuple(_ -> 0f0, 9), ntuple(_ -> 0f0, 16)

function envmap_index!(fl::Flattener, em)
    D = em.distribution                                                # sampler/sampling.jl:179-262, stored verbatim
    data = Array{Hikari.RGBSpectrum}(em.data)                          # Matrix{RGBSpectrum}[h, w]
    cf, cc = Array{Float32}(D.conditional_func), Array{Float32}(D.conditional_cdf)
    cfi, mf, mc = Vector{Float32}(D.conditional_func_int), Vector{Float32}(D.marginal_func), Vector{Float32}(D.marginal_cdf)
    push!(fl.keep, data, cf, cc, cfi, mf, mc)
    h, w = size(data)
    push!(fl.envmaps, HkEnvmap(w, h, Ptr{Float32}(pointer(data)), rowmajor3(em.rotation), D.nu, D.nv, pointer(cf), pointer(cc), pointer(cfi), pointer(mf), pointer(mc),
                               D.marginal_func_int, 0))
    Int32(length(fl.envmaps) - 1)
end

function light_record(fl::Flattener, l)
    if l isa Hikari.DiffuseAreaLight                                   # lights/diffuse-area.jl:25-33, one per emissive face (scene-mesh.jl:98-131)
        v = l.vertices
        return HkLight(HK_LIGHT_DIFFUSE_AREA, HK_SPEC_RGB, (0f0, 0f0, 0f0, 1f0), Z3, 0f0, l.scale, Z3, Z3, Z16, Z16, 0f0, 0f0,
                       (v[1][1], v[1][2], v[1][3], v[2][1], v[2][2], v[2][3], v[3][1], v[3][2], v[3][3]), tup3(l.normal), l.area,
                       (l.uv[1][1], l.uv[1][2], l.uv[2][1], l.uv[2][2], l.uv[3][1], l.uv[3][2]), tex_rgba(fl, l.Le), l.two_sided, Int32(-1))
    elseif l isa Hikari.EnvironmentLight                               # lights/environment.jl:5-19: scale::RGBSpectrum rides in i_rgb
        return HkLight(HK_LIGHT_ENVIRONMENT, HK_SPEC_RGB, rgba(l.scale), Z3, 0f0, 1f0, Z3, Z3, Z16, Z16, 0f0, 0f0, Z9, Z3, 0f0, Z6, NO_RGBA, 0, envmap_index!(fl, l.env_map))
    end
    sk, irgb, poly, isc = spectrum_fields(l.i)
    if l isa Hikari.PointLight
        return HkLight(HK_LIGHT_POINT, sk, irgb, poly, isc, l.scale, tup3(l.position), Z3, Z16, Z16, 0f0, 0f0, Z9, Z3, 0f0, Z6, NO_RGBA, 0, Int32(-1))
    elseif l isa Hikari.SpotLight
        return HkLight(HK_LIGHT_SPOT, sk, irgb, poly, isc, l.scale, tup3(l.position), Z3, rowmajor(l.world_to_light.m), rowmajor(l.light_to_world.m),
                       l.cos_total_width, l.cos_falloff_start, Z9, Z3, 0f0, Z6, NO_RGBA, 0, Int32(-1))
    elseif l isa Hikari.SunLight
        return HkLight(HK_LIGHT_SUN, sk, irgb, poly, isc, l.scale, Z3, tup3(l.direction), Z16, Z16, 0f0, 0f0, Z9, Z3, 0f0, Z6, NO_RGBA, 0, Int32(-1))
    elseif l isa Hikari.DirectionalLight
        return HkLight(HK_LIGHT_DIRECTIONAL, sk, irgb, poly, isc, l.scale, Z3, tup3(l.direction), Z16, Z16, 0f0, 0f0, Z9, Z3, 0f0, Z6, NO_RGBA, 0, Int32(-1))
    elseif l isa Hikari.AmbientLight
        return HkLight(HK_LIGHT_AMBIENT, sk, irgb, poly, isc, l.scale, Z3, Z3, Z16, Z16, 0f0, 0f0, Z9, Z3, 0f0, Z6, NO_RGBA, 0, Int32(-1))
    end
    error("HikariMI355X: light type $(typeof(l)) is not part of the VolPath path")
end

i3(v) = (Int32(v[1]), Int32(v[2]), Int32(v[3]))
const I3Z = (Int32(0), Int32(0), Int32(0))
function grid_ptr!(fl::Flattener, g)
    g === nothing && return Ptr{Float32}(C_NULL)
    arr = Array(g)
    push!(fl.keep, arr)
    Ptr{Float32}(pointer(arr))
end
function medium_record(fl::Flattener, m)
    nul, nul8 = Ptr{Float32}(C_NULL), Ptr{UInt8}(C_NULL)
    z4 = (0f0, 0f0, 0f0, 1f0)
    if m isa Hikari.HomogeneousMedium                                  # volpath/media.jl:762-776
        return HkMedium(HK_MEDIUM_HOMOGENEOUS, rgba(m.σ_a), rgba(m.σ_s), rgba(m.Le), m.g, 1f0, 1f0, Z3, Z3, Z16, Z16, I3Z, nul, nul, nul, nul, I3Z, nul, 0f0,
                        nul8, 0, 0, 0, 0, 0, 0, 0, 0, 0, Z9, Z3, I3Z, I3Z)
    elseif m isa Hikari.GridMedium                                     # :873-935; density [nx,ny,nz] x fastest; majorant x + rx*(y + ry*z)
        mg = m.majorant_grid
        return HkMedium(HK_MEDIUM_GRID, rgba(m.σ_a), rgba(m.σ_s), z4, m.g, 1f0, 1f0, tup3(m.bounds.p_min), tup3(m.bounds.p_max), rowmajor(m.render_to_medium),
                        rowmajor(m.medium_to_render), i3(m.density_res), grid_ptr!(fl, m.density), nul, nul, nul, i3(mg.res), grid_ptr!(fl, mg.voxels), m.max_density,
                        nul8, 0, 0, 0, 0, 0, 0, 0, 0, 0, Z9, Z3, I3Z, I3Z)
    elseif m isa Hikari.RGBGridMedium                                  # :1002-1113; RGBSpectrum voxels = 4 floats each
        mg = m.majorant_grid
        return HkMedium(HK_MEDIUM_RGB_GRID, z4, z4, z4, m.g, m.sigma_scale, m.Le_scale, tup3(m.bounds.p_min), tup3(m.bounds.p_max), rowmajor(m.render_to_medium),
                        rowmajor(m.medium_to_render), i3(m.grid_res), nul, grid_ptr!(fl, m.σ_a_grid), grid_ptr!(fl, m.σ_s_grid), grid_ptr!(fl, m.Le_grid), i3(mg.res),
                        grid_ptr!(fl, mg.voxels), 0f0, nul8, 0, 0, 0, 0, 0, 0, 0, 0, 0, Z9, Z3, I3Z, I3Z)
    elseif m isa Hikari.NanoVDBMedium                                  # nanovdb.jl:153-191: the raw grid bytes and the reference's 1-based offsets
        mg = m.majorant_grid
        buf = Vector{UInt8}(m.buffer)
        push!(fl.keep, buf)
        return HkMedium(HK_MEDIUM_NANOVDB, rgba(m.σ_a), rgba(m.σ_s), z4, m.g, 1f0, 1f0, tup3(m.bounds.p_min), tup3(m.bounds.p_max), Z16, Z16, I3Z, nul, nul, nul, nul,
                        i3(mg.res), grid_ptr!(fl, mg.voxels), m.max_density, pointer(buf), length(buf), m.root_offset, m.upper_offset, m.lower_offset, m.leaf_offset,
                        m.upper_count, m.lower_count, m.leaf_count, m.root_table_size, m.inv_mat, m.vec, m.index_bbox_min, m.index_bbox_max)
    end
    error("HikariMI355X: medium type $(typeof(m)) is not part of the VolPath path")
end

"""
    flatten_scene(ctx, scene) -> hk_scene

World-space triangle soup + TriangleMeta from the TLAS, materials / media by MultiTypeSet slot, lights in the flat order of
`flat_to_light_index` (lights/light-sampler.jl:289-329), medium interfaces; then `hk_scene_create` (BVH + light BVH are built by
the library).  Arrays are borrowed for the call only.
"""
function flatten_scene(ctx::Ptr{Cvoid}, scene)
    fl = Flattener(Raycore.get_static(scene.materials), HkTexture[], Dict{UInt,Int32}(), HkPlSpectrum[], HkEnvmap[], Any[])
    # ---- materials / media: flat index = position in slot order; SetKey -> flat index ----
    mat_base, n = Int32[], Int32(0)
    for vec in scene.materials
        push!(mat_base, n); n += Int32(length(vec))
    end
    flat_mat(k) = Raycore.is_valid(k) ? mat_base[k.type_idx] + Int32(k.vec_idx) - Int32(1) : Int32(-1)
    materials = HkMaterial[]
    for vec in scene.materials, m in vec
        push!(materials, material_record(fl, m, flat_mat))
    end
    med_base, n = Int32[], Int32(0)
    for vec in scene.media
        push!(med_base, n); n += Int32(length(vec))
    end
    flat_med(k) = Raycore.is_valid(k) ? med_base[k.type_idx] + Int32(k.vec_idx) - Int32(1) : Int32(-1)
    media = HkMedium[]
    for vec in scene.media, m in vec
        push!(media, medium_record(fl, m))
    end
    mis = [HkMediumInterface(flat_mat(mi.material), flat_med(mi.inside), flat_med(mi.outside)) for mi in scene.media_interfaces]
    # ---- lights: type slots in order == the flat light index the TriangleMeta and the light sampler use ----
    lights = HkLight[]
    fl.static_mats = Raycore.get_static(scene.lights)                  # a textured DiffuseAreaLight.Le refers to the LIGHTS set's textures
    for vec in scene.lights, l in vec
        push!(lights, light_record(fl, l))
    end
    # ---- geometry: every TLAS instance's triangles, transformed to world space ----
    pos, nrm, uvs, tan, meta = Float32[], Float32[], Float32[], Float32[], HkTriMeta[]
    accel = scene.accel
    for inst in accel.instances
        blas = instance_blas(accel, inst)
        m, im = inst.transform, inst.inv_transform
        for tri in blas.primitives
            vs, ns, ts, uv = Raycore.vertices(tri), Raycore.normals(tri), Raycore.tangents(tri), Raycore.uvs(tri)
            for k in 1:3
                append!(pos, xform_point(m, vs[k]))
                append!(nrm, xform_normal(im, ns[k]))
                append!(tan, xform_dir(m, ts[k]))
                push!(uvs, Float32(uv[k][1]), Float32(uv[k][2]))
            end
            tm = tri.metadata::Hikari.TriangleMeta                     # scene.jl:11-15: interface index is 1-based on the Julia side
            push!(meta, HkTriMeta(tm.medium_interface_idx - UInt32(1), tm.primitive_index, tm.arealight_flat_idx))
        end
    end
    textures, spectra, envmaps = fl.textures, fl.spectra, fl.envmaps
    out = Ref{Ptr{Cvoid}}()
    keep = fl.keep
    GC.@preserve keep pos nrm uvs tan meta materials textures mis lights envmaps media spectra begin
        p(a) = isempty(a) ? Ptr{eltype(a)}(C_NULL) : pointer(a)
        desc = HkSceneDesc(length(meta), length(materials), length(textures), length(mis), length(lights), length(envmaps), length(media), length(spectra),
                           p(pos), p(nrm), p(uvs), p(tan), p(meta), p(materials), p(textures), p(mis), p(lights), p(envmaps), p(media), p(spectra))
        check(ccall((:hk_scene_create, LIB), Int32, (Ptr{Cvoid}, Ref{HkSceneDesc}, Ref{Ptr{Cvoid}}), ctx, desc, out), "hk_scene_create")
    end
    out[]
end

# ---------------------------------------------------------------------------------------------------------------------------
# Cameras: one record covers PerspectiveCamera (camera/perspective.jl:41-128) and MatrixCamera (camera/matrix.jl:13-115, no lens)
# ---------------------------------------------------------------------------------------------------------------------------
function camera_record(cam::Hikari.PerspectiveCamera)
    HkCamera(rowmajor(cam.core.raster_to_camera.m), rowmajor(cam.core.core.camera_to_world.m), cam.core.lens_radius, cam.core.focal_distance,
             cam.core.core.shutter_open, cam.core.core.shutter_close, tup3(cam.dx_camera), tup3(cam.dy_camera))
end
function camera_record(cam::Hikari.MatrixCamera)
    HkCamera(rowmajor(cam.raster_to_camera.m), rowmajor(cam.core.camera_to_world.m), 0f0, 1f6, cam.core.shutter_open, cam.core.shutter_close,
             tup3(cam.dx_camera), tup3(cam.dy_camera))
end

# ---------------------------------------------------------------------------------------------------------------------------
# render! / functor / clear! / close  (volpath.jl:445-670, 108-113; Hikari.jl:47)
# ---------------------------------------------------------------------------------------------------------------------------
function render_samples!(vp::MI355XVolPath, scene, film::Hikari.Film, camera, n::Int; display::Symbol = :every)
    ensure_ctx!(vp)
    h, w = size(film.framebuffer)
    G = length(vp.devs)
    if vp.devs[1].film == C_NULL || vp.film_size != (w, h)
        vp.pinned_fb = nothing                              # (hk_film_destroy unregisters)
        for d in vp.devs
            d.film != C_NULL && ccall((:hk_film_destroy, LIB), Int32, (Ptr{Cvoid},), d.film)
            r = Ref{Ptr{Cvoid}}()
            check(ccall((:hk_film_create, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), d.ctx, w, h, vp.params.accumulate_f64, C_NULL, r), "hk_film_create")
            d.film = r[]
            check(ccall((:hk_film_clear, LIB), Int32, (Ptr{Cvoid},), d.film), "hk_film_clear")
        end
        vp.film_size = (w, h)
    end
    if vp.devs[1].scene == C_NULL || vp.scene_id != objectid(scene)
        for d in vp.devs
            d.scene != C_NULL && ccall((:hk_scene_destroy, LIB), Int32, (Ptr{Cvoid},), d.scene)
            d.scene = flatten_scene(d.ctx, scene)                       # the scene is replicated on every device (SURVEY 8e)
        end
        vp.scene_id = objectid(scene)
    end
    first = film.iteration_index[] + Int32(1)
    cam = camera_record(camera)
    # sample-index sharding: device g renders first+g, first+g+G, ... ; every hk_render only enqueues work on its context's stream
    for (g, d) in enumerate(vp.devs)
        cnt = n >= g ? cld(n - (g - 1), G) : 0
        cnt == 0 && continue
        check(ccall((:hk_render, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{HkCamera}, Int32, Int32, Int32),
                    d.ctx, d.scene, d.integ, d.film, cam, first + Int32(g - 1), cnt, G), "hk_render")
    end
    film.iteration_index[] = first + Int32(n - 1)
    root = vp.devs[1]
    if G > 1
        # The film of a multi-device frame is the SUM of the per-device accumulators.  hk_film_reduce adds them on the root IN PLACE,
        # so the other devices restart from zero afterwards (their samples now live in the root's accumulators).
        films = [d.film for d in vp.devs]
        GC.@preserve films check(ccall((:hk_film_reduce, LIB), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Int32, Int32), vp.comm, pointer(films), G, 0), "hk_film_reduce")
        for d in vp.devs[2:end]
            check(ccall((:hk_sync, LIB), Int32, (Ptr{Cvoid},), d.ctx), "hk_sync")
            check(ccall((:hk_film_clear, LIB), Int32, (Ptr{Cvoid},), d.film), "hk_film_clear")
        end
    end
    fb = film.framebuffer                               # Matrix{RGB{Float32}}[h, w]: exactly hk_film_read_rgb's layout
    if display === :every
        vp.read_pending = false
        if vp.pinned_fb !== fb          # the viewer's one framebuffer: the frame is copied straight into it (a refusal only costs the memcpy)
            ok = GC.@preserve fb ccall((:hk_film_pin_host, LIB), Int32, (Ptr{Cvoid}, Ptr{Float32}), root.film, Ptr{Float32}(pointer(fb)))
            vp.pinned_fb = ok == 0 ? fb : nothing
        end
        GC.@preserve fb check(ccall((:hk_film_read_rgb, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}), root.ctx, root.film, Ptr{Float32}(pointer(fb))), "hk_film_read_rgb")
    elseif display === :pipelined
        # this call's samples go to the GPU now; the frame of the PREVIOUS call is collected while they render; then this call's copy is enqueued
        check(ccall((:hk_flush, LIB), Int32, (Ptr{Cvoid},), root.ctx), "hk_flush")
        if vp.read_pending
            GC.@preserve fb check(ccall((:hk_film_read_wait, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}, Ptr{Ptr{Float32}}), root.ctx, root.film, Ptr{Float32}(pointer(fb)), C_NULL), "hk_film_read_wait")
        end
        check(ccall((:hk_film_read_rgb_async, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), root.ctx, root.film), "hk_film_read_rgb_async")
        vp.read_pending = true
    end                                                 # :skip — the call is only noted (hk_render's ordering contract); the next read-back renders it
    nothing
end

"film.framebuffer <- the frame of everything rendered so far (after `display = :pipelined` / `:manual` loops, or between displayed calls)"
function sync_display!(vp::MI355XVolPath, film::Hikari.Film)
    root = vp.devs[1]
    (root.ctx == C_NULL || root.film == C_NULL) && return nothing
    fb = film.framebuffer
    vp.read_pending = false
    vp.calls_since_display = 0
    GC.@preserve fb check(ccall((:hk_film_read_rgb, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}), root.ctx, root.film, Ptr{Float32}(pointer(fb))), "hk_film_read_rgb")
    nothing
end

function Hikari.render!(vp::MI355XVolPath, scene::Hikari.AbstractScene, film::Hikari.Film, camera::Hikari.Camera)
    vp.calls_since_display += 1
    due = vp.display !== :manual && vp.calls_since_display >= vp.display_every
    due && (vp.calls_since_display = 0)
    # multi-device frames reduce (and look) in every call: batching is a one-device matter
    render_samples!(vp, scene, film, camera, 1; display = !due && length(vp.devs) == 1 ? :skip : (vp.display === :pipelined && due ? :pipelined : :every))
end

function Hikari.clear!(vp::MI355XVolPath)
    for d in vp.devs
        d.film != C_NULL && check(ccall((:hk_film_clear, LIB), Int32, (Ptr{Cvoid},), d.film), "hk_film_clear")
    end
end

function (vp::MI355XVolPath)(scene::Hikari.AbstractScene, film::Hikari.Film, camera::Hikari.Camera)
    film.iteration_index[] = Int32(0)
    Hikari.clear!(vp)
    vp.calls_since_display = 0
    render_samples!(vp, scene, film, camera, Int(vp.samples_per_pixel))
    copyto!(film.postprocess, film.framebuffer)          # the functor returns film.postprocess (volpath.jl:669): linear HDR until postprocess! runs
    return film.postprocess
end

function Base.close(vp::MI355XVolPath)
    vp.read_pending = false
    vp.devs[1].film != C_NULL && vp.pinned_fb !== nothing && ccall((:hk_film_unpin_host, LIB), Int32, (Ptr{Cvoid},), vp.devs[1].film)
    vp.pinned_fb = nothing
    vp.comm != C_NULL && ccall((:hk_comm_destroy, LIB), Int32, (Ptr{Cvoid},), vp.comm)
    vp.comm = C_NULL
    for d in vp.devs
        d.film != C_NULL && ccall((:hk_film_destroy, LIB), Int32, (Ptr{Cvoid},), d.film)
        d.integ != C_NULL && ccall((:hk_integrator_destroy, LIB), Int32, (Ptr{Cvoid},), d.integ)
        d.scene != C_NULL && ccall((:hk_scene_destroy, LIB), Int32, (Ptr{Cvoid},), d.scene)
        d.ctx != C_NULL && ccall((:hk_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), d.ctx)
        d.film = d.integ = d.scene = d.ctx = C_NULL
    end
    nothing
end

# ---------------------------------------------------------------------------------------------------------------------------
# postprocess! / denoise! / fill_aux_buffers! forwards (postprocess.jl:293-357, denoise.jl:301-376, film.jl:410-483)
# ---------------------------------------------------------------------------------------------------------------------------
const TONEMAPS = Dict(nothing => Int32(0), :none => Int32(0), :reinhard => Int32(1), :reinhard_extended => Int32(2), :aces => Int32(3), :uncharted2 => Int32(4), :filmic => Int32(5))

"""
    postprocess!(vp, film; exposure=1f0, tonemap=:aces, gamma=2.2f0, white_point=4f0, sensor=nothing, background=nothing)

The keyword set of `Hikari.postprocess!`; symbols / `FilmSensor` / the Bradford matrix are resolved here into the kernel arguments
of `postprocess_kernel!` (postprocess.jl:185-250) and `hk_postprocess` runs it on `film.framebuffer` into `film.postprocess`.
"""
function postprocess!(vp::MI355XVolPath, film::Hikari.Film; exposure::Real = 1f0, tonemap = :aces, gamma = 2.2f0, white_point::Real = 4f0,
                      sensor = nothing, background = nothing)
    ensure_ctx!(vp)
    h, w = size(film.framebuffer)
    ratio, apply_wb, wb = 1f0, Int32(0), ntuple(i -> i in (1, 5, 9) ? 1f0 : 0f0, 9)
    if sensor !== nothing
        ratio = Float32(sensor.exposure_time * sensor.iso / 100f0)
        if sensor.white_balance > 0
            apply_wb, wb = Int32(1), rowmajor3(Hikari.compute_white_balance_matrix(Float32(sensor.white_balance)))   # spectral/color.jl:522-547
        end
    end
    mask, bg = background === nothing ? (Int32(0), Z3) : (Int32(1), (Float32(background.r), Float32(background.g), Float32(background.b)))
    P = HkPostprocessParams(Float32(exposure), TONEMAPS[tonemap], gamma === nothing ? 1f0 : 1f0 / Float32(gamma), gamma === nothing ? 0 : 1, Float32(white_point),
                            ratio, apply_wb, wb, mask, bg)
    src, dst, depth = film.framebuffer, film.postprocess, film.depth
    GC.@preserve src dst depth check(ccall((:hk_postprocess, LIB), Int32, (Ptr{Cvoid}, Ref{HkPostprocessParams}, Int32, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                                           vp.devs[1].ctx, P, w, h, Ptr{Float32}(pointer(src)), mask == 1 ? pointer(depth) : Ptr{Float32}(C_NULL), Ptr{Float32}(pointer(dst))),
                                     "hk_postprocess")
    film.postprocess
end

"denoise!(film; config) (denoise.jl:301-376): à-trous passes guided by film.normal / film.depth; even passes write into film.framebuffer like the reference."
function denoise!(vp::MI355XVolPath, film::Hikari.Film; config = Hikari.DenoiseConfig())
    ensure_ctx!(vp)
    h, w = size(film.framebuffer)
    P = HkDenoiseParams(config.iterations, config.sigma_color, config.sigma_normal, config.sigma_depth, config.use_variance)
    src, nrm, dep, dst = film.framebuffer, film.normal, film.depth, film.postprocess
    GC.@preserve src nrm dep dst check(ccall((:hk_denoise, LIB), Int32,
                                             (Ptr{Cvoid}, Ref{HkDenoiseParams}, Int32, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                                             vp.devs[1].ctx, P, w, h, Ptr{Float32}(pointer(src)), Ptr{Float32}(pointer(nrm)), pointer(dep), Ptr{Float32}(pointer(dst)),
                                             Ptr{Float32}(pointer(src))), "hk_denoise")
    film.postprocess
end

"fill_aux_buffers!(film, scene, camera; has_infinite_lights) (film.jl:410-483): first-hit albedo / normal / depth per pixel centre."
function fill_aux_buffers!(vp::MI355XVolPath, film::Hikari.Film, scene, camera; has_infinite_lights::Bool = false)
    ensure_ctx!(vp)
    d = vp.devs[1]
    if d.scene == C_NULL || vp.scene_id != objectid(scene)
        d.scene != C_NULL && ccall((:hk_scene_destroy, LIB), Int32, (Ptr{Cvoid},), d.scene)
        d.scene = flatten_scene(d.ctx, scene)
        vp.scene_id = length(vp.devs) == 1 ? objectid(scene) : UInt(0)
    end
    h, w = size(film.framebuffer)
    cam = camera_record(camera)
    alb, nrm, dep = film.albedo, film.normal, film.depth
    GC.@preserve alb nrm dep check(ccall((:hk_film_fill_aux, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{HkCamera}, Int32, Int32, Int32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                                         d.ctx, d.scene, cam, w, h, has_infinite_lights, Ptr{Float32}(pointer(alb)), Ptr{Float32}(pointer(nrm)), pointer(dep)), "hk_film_fill_aux")
    nothing
end

end # module
