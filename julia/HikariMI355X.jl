# HikariMI355X.jl — the reference-side binding a Hikari.jl maintainer would add (see INTEGRATION.md).
# NOT exercised in this repository: the build image has no Julia toolchain (SURVEY.md, "Facts established").
# It subtypes Hikari.Integrator and forwards the VolPath hot path to libhikari_mi355x.so via ccall.
module HikariMI355X

using Hikari
import Hikari: Integrator, render!, clear!
using Raycore

const LIB = get(ENV, "HIKARI_MI355X_LIB", "libhikari_mi355x")

check(st::Int32, what) = st == 0 || error("$what failed ($st): " * unsafe_string(ccall((:hk_last_error, LIB), Cstring, ())))

# ---- POD mirrors of include/hikari_mi355x.h (field order must match) --------------------------------
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
struct HkTables
    sobol::Ptr{UInt32}; sobol_count::Int32; rgb2spec_res::Int32
    cie_x::Ptr{Float32}; cie_y::Ptr{Float32}; cie_z::Ptr{Float32}
    rgb2spec_scale::Ptr{Float32}; rgb2spec_coeffs::Ptr{Float32}
end

mutable struct MI355XVolPath <: Integrator
    params::HkIntegratorParams
    samples_per_pixel::Int32
    devices::Vector{Int}
    ctx::Ptr{Cvoid}; integ::Ptr{Cvoid}; film::Ptr{Cvoid}; scene::Ptr{Cvoid}
    scene_id::UInt
    film_size::Tuple{Int,Int}
end

rowmajor(m) = ntuple(i -> Float32(m[(i - 1) ÷ 4 + 1, (i - 1) % 4 + 1]), 16)

function MI355XVolPath(; max_depth::Int = 8, samples::Int = 64, russian_roulette_depth::Int = 3, regularize::Bool = true,
                       material_coherence::Symbol = :none, max_component_value::Real = 10f0,
                       filter::Hikari.AbstractFilter = Hikari.GaussianFilter(), accumulation_eltype::DataType = Float32,
                       devices = 0:0)
    @assert material_coherence in (:none, :sorted, :per_type)
    @assert accumulation_eltype in (Float32, Float64)
    fp = Hikari.GPUFilterParams(filter)
    p = HkIntegratorParams(max_depth, samples, russian_roulette_depth, regularize, findfirst(==(material_coherence), (:none, :sorted, :per_type)) - 1,
                           Float32(max_component_value), fp.filter_type, (fp.radius[1], fp.radius[2]), fp.param1, fp.param2,
                           accumulation_eltype === Float64, UInt32(0), 0)
    MI355XVolPath(p, samples, collect(devices), C_NULL, C_NULL, C_NULL, C_NULL, UInt(0), (0, 0))
end

function ensure_ctx!(vp::MI355XVolPath)
    vp.ctx != C_NULL && return
    r = Ref{Ptr{Cvoid}}()
    check(ccall((:hk_ctx_create, LIB), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), vp.devices[1], C_NULL, r), "hk_ctx_create")
    vp.ctx = r[]
    tab = Hikari.get_srgb_table()
    sob, cx, cy, cz = Hikari.SobolMatrices32, Hikari.CIE_X, Hikari.CIE_Y, Hikari.CIE_Z
    GC.@preserve tab sob cx cy cz begin
        t = HkTables(pointer(sob), length(sob), tab.res, pointer(cx), pointer(cy), pointer(cz), pointer(tab.scale), pointer(tab.coeffs))
        check(ccall((:hk_ctx_set_tables, LIB), Int32, (Ptr{Cvoid}, Ref{HkTables}), vp.ctx, t), "hk_ctx_set_tables")
    end
    ri = Ref{Ptr{Cvoid}}()
    check(ccall((:hk_integrator_create, LIB), Int32, (Ptr{Cvoid}, Ref{HkIntegratorParams}, Ref{Ptr{Cvoid}}), vp.ctx, vp.params, ri), "hk_integrator_create")
    vp.integ = ri[]
end

# flatten_scene(scene) walks scene.accel (TLAS instances -> world-space triangles + TriangleMeta), scene.materials,
# scene.lights (flat order of Hikari.flat_to_light_index), scene.media_interfaces and fills hk_scene_desc;
# see hikari.jl_amd/scene.py::Scene.sync for the exact record-by-record mapping (same field names).
function flatten_scene end

function camera_record(cam::Hikari.PerspectiveCamera)
    HkCamera(rowmajor(cam.core.raster_to_camera.m), rowmajor(cam.core.core.camera_to_world.m), cam.core.lens_radius, cam.core.focal_distance,
             cam.core.core.shutter_open, cam.core.core.shutter_close, Tuple(cam.dx_camera), Tuple(cam.dy_camera))
end

function render_samples!(vp::MI355XVolPath, scene, film::Hikari.Film, camera, n::Int)
    ensure_ctx!(vp)
    h, w = size(film.framebuffer)
    if vp.film == C_NULL || vp.film_size != (w, h)
        r = Ref{Ptr{Cvoid}}()
        check(ccall((:hk_film_create, LIB), Int32, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), vp.ctx, w, h, vp.params.accumulate_f64, C_NULL, r), "hk_film_create")
        vp.film, vp.film_size = r[], (w, h)
    end
    if vp.scene == C_NULL || vp.scene_id != objectid(scene)
        vp.scene = flatten_scene(vp, scene)          # hk_scene_create inside
        vp.scene_id = objectid(scene)
    end
    first = film.iteration_index[] + Int32(1)
    cam = camera_record(camera)
    check(ccall((:hk_render, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{HkCamera}, Int32, Int32, Int32),
                vp.ctx, vp.scene, vp.integ, vp.film, cam, first, n, 1), "hk_render")
    film.iteration_index[] = first + Int32(n - 1)
    fb = film.framebuffer                               # Matrix{RGB{Float32}}[h, w]: exactly hk_film_read_rgb's layout
    GC.@preserve fb check(ccall((:hk_film_read_rgb, LIB), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float32}), vp.ctx, vp.film, pointer(fb)), "hk_film_read_rgb")
    nothing
end

Hikari.render!(vp::MI355XVolPath, scene::Hikari.AbstractScene, film::Hikari.Film, camera::Hikari.Camera) = render_samples!(vp, scene, film, camera, 1)

function Hikari.clear!(vp::MI355XVolPath)
    vp.film != C_NULL && check(ccall((:hk_film_clear, LIB), Int32, (Ptr{Cvoid},), vp.film), "hk_film_clear")
end

function (vp::MI355XVolPath)(scene::Hikari.AbstractScene, film::Hikari.Film, camera::Hikari.Camera)
    film.iteration_index[] = Int32(0)
    Hikari.clear!(vp)
    render_samples!(vp, scene, film, camera, Int(vp.samples_per_pixel))
    return film.postprocess
end

function Base.close(vp::MI355XVolPath)
    vp.film != C_NULL && ccall((:hk_film_destroy, LIB), Int32, (Ptr{Cvoid},), vp.film)
    vp.integ != C_NULL && ccall((:hk_integrator_destroy, LIB), Int32, (Ptr{Cvoid},), vp.integ)
    vp.scene != C_NULL && ccall((:hk_scene_destroy, LIB), Int32, (Ptr{Cvoid},), vp.scene)
    vp.ctx != C_NULL && ccall((:hk_ctx_destroy, LIB), Int32, (Ptr{Cvoid},), vp.ctx)
    vp.film = vp.integ = vp.scene = vp.ctx = C_NULL
    nothing
end

end # module
