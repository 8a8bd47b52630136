# julia/make_reference_fixtures.jl — ONE command that pins the oracle to Hikari itself, for the first box that has Julia + Hikari.jl
# (with its Raycore branch).  No GPU needed.
#
#     julia --project=<env with Hikari> julia/make_reference_fixtures.jl            # writes tests/golden/reference/
#     python -m pytest tests/test_reference_fixtures.py                               # oracle (CPU) and, with -m gpu, the device against them
#
# It has NEVER been executed (no Julia in the build image).  tests/test_reference_fixtures.py checks it statically: every `Hikari.`
# function named in a `# ref:` comment below exists at the cited file:line of the reference tree, and the arrays it reads / writes
# are the ones the Python side consumes.  Until it has run, tests/golden/reference/ is empty and those tests SKIP with a loud reason;
# the oracle stays "parity unpinned" (DESIGN.md section 2).
#
# What it does: reads the committed inputs (tests/golden/reference_inputs/, written by tests/golden/make_reference_inputs.py), calls
# Hikari's OWN per-stage functions on them — nothing is restated here but the few lines of glue each kernel wraps around them — and
# dumps the results in the raw format of tests/fixture_io.py (`<name>.bin` + manifest.txt; a NumPy [n, k] array is a Julia k x n Matrix).
#
# Stages (the SURVEY 8 rows they pin):
#   table     a22  the RGB -> spectrum table this Julia uses (srgb_spectrum_table.dat layout): the Python side loads IT for every stage below
#   sobol     a8   zsobol_sample_1d / zsobol_sample_2d
#   camera    a6 a9 a10 a21  K1 per filter kind: compute_pixel_sample, filter_sample (filter_sample_tabulated inside), sample_wavelengths_visible, generate_ray
#   uplift    a22  uplift_rgb / uplift_rgb_unbounded / uplift_rgb_illuminant
#   bsdf      a17 a18 a19  sample_bsdf_spectral / evaluate_bsdf_spectral of nine material kinds (+ Matte sigma, smooth conductor, measured gold)
#   lightbvh  a25 a26  BVHLightSampler + bvh_sample_light / bvh_pmf + the node array
#   light     a24  sample_light_spectral per light kind (envlight: an EnvironmentLight over a 16 x 16 map, with its Distribution2D's cdfs)
#   nanovdb   a28 a29  build_nanovdb_from_dense + nanovdb_get_value + sample_point (trilinear) + the majorant grid
#   frame     a2   the 64 x 64 scene of test/volpath_integration.jl through Hikari.VolPath (surfaces only: per pixel; with fog: means)
using Hikari
using GeometryBasics
using GeometryBasics: normal_mesh, Tesselation, Point3f, Vec3f, Point2f
using StaticArrays
import KernelAbstractions as KA
import Raycore

const ROOT = normpath(joinpath(@__DIR__, ".."))
const IN_DIR = joinpath(ROOT, "tests", "golden", "reference_inputs")
const OUT_DIR = joinpath(ROOT, "tests", "golden", "reference")
const DT = Dict("f32" => Float32, "i32" => Int32, "u32" => UInt32, "u8" => UInt8)
const DN = Dict(Float32 => "f32", Int32 => "i32", UInt32 => "u32", UInt8 => "u8")

# ---- raw fixture sets (tests/fixture_io.py) -----------------------------------------------------------------------------
function read_set(dir)
    out = Dict{String,Array}()
    for line in eachline(joinpath(dir, "manifest.txt"))
        parts = split(line)
        isempty(parts) && continue
        T = DT[parts[2]]
        dims = reverse(parse.(Int, parts[3:end]))            # NumPy (row-major) dims -> Julia (column-major) dims
        a = Array{T}(undef, dims...)
        read!(joinpath(dir, parts[1] * ".bin"), a)
        out[String(parts[1])] = a
    end
    out
end
const OUT = Dict{String,Array}()
put!(name, a::Array) = (OUT[name] = a)
function write_set(dir, arrays)
    mkpath(dir)
    open(joinpath(dir, "manifest.txt"), "w") do mf
        for name in sort(collect(keys(arrays)))
            a = arrays[name]
            write(joinpath(dir, name * ".bin"), a)
            println(mf, join([name, DN[eltype(a)], string.(reverse(size(a)))...], " "))
        end
    end
end

const IN = read_set(IN_DIR)
const N = size(IN["bsdf_wo"], 2)
v3(a, i) = Vec3f(a[1, i], a[2, i], a[3, i])
p3(a, i) = Point3f(a[1, i], a[2, i], a[3, i])
p2(a, i) = Point2f(a[1, i], a[2, i])
# Wavelengths with the pdf of the visible-wavelength sampler (what every path carries)                    # ref: Hikari.visible_wavelengths_pdf spectral/spectral.jl:192
wl(a, i) = Hikari.Wavelengths((a[1, i], a[2, i], a[3, i], a[4, i]),
                              ntuple(k -> Hikari.visible_wavelengths_pdf(a[k, i]), 4))
spec4(s) = (s[1], s[2], s[3], s[4])

# ---- table ---------------------------------------------------------------------------------------------------------------
const TABLE = Hikari.get_srgb_table()                                                                      # ref: Hikari.get_srgb_table spectral/rgb2spec.jl:424
Hikari.save_srgb_table_binary(joinpath(OUT_DIR * "_table.tmp"), TABLE)                                     # ref: Hikari.save_srgb_table_binary spectral/rgb2spec.jl:415

# ---- sobol ---------------------------------------------------------------------------------------------------------------
function stage_sobol()
    log2_spp, n_digits = Hikari.compute_zsobol_params(4096, 64, 64)                                        # ref: Hikari.compute_zsobol_params sampler/sobol.jl:317
    px, py, si, dm = IN["sobol_px"], IN["sobol_py"], IN["sobol_sidx"], IN["sobol_dim"]
    o1 = Vector{Float32}(undef, length(px))
    o2 = Matrix{Float32}(undef, 2, length(px))
    for i in eachindex(px)
        o1[i] = Hikari.zsobol_sample_1d(px[i], py[i], si[i], dm[i], log2_spp, n_digits, UInt32(0), Hikari.SobolMatrices32)   # ref: Hikari.zsobol_sample_1d sampler/sobol.jl:269
        a, b = Hikari.zsobol_sample_2d(px[i], py[i], si[i], dm[i], log2_spp, n_digits, UInt32(0), Hikari.SobolMatrices32)    # ref: Hikari.zsobol_sample_2d sampler/sobol.jl:290
        o2[1, i], o2[2, i] = a, b
    end
    put!("sobol_1d", o1)
    put!("sobol_2d", o2)
end

# ---- camera (K1, volpath.jl:143-181: the same calls in the same order) -----------------------------------------------------
const FILTERS = [("box", Hikari.BoxFilter(Point2f(0.5f0, 0.5f0))), ("triangle", Hikari.TriangleFilter(Point2f(2f0, 2f0))),
                 ("gaussian", Hikari.GaussianFilter(Point2f(1.5f0, 1.5f0), 0.5f0)), ("mitchell", Hikari.MitchellFilter(Point2f(2f0, 2f0), 1f0 / 3f0, 1f0 / 3f0)),
                 ("lanczos", Hikari.LanczosSincFilter(Point2f(4f0, 4f0), 3f0))]
function stage_camera()
    film = Hikari.Film(Point2f(64, 64))
    camera = Hikari.PerspectiveCamera(Point3f(0f0, 1f0, -3.5f0), Point3f(0f0, 1f0, 0f0), film; fov=40f0)
    rng = Hikari.SobolRNG(KA.CPU(), UInt32(0), 64, 64, 4096)                                               # ref: Hikari.SobolRNG sampler/sobol.jl:370
    px, py, si = IN["cam_px"], IN["cam_py"], IN["cam_sidx"]
    height = Int32(64)
    for (name, f) in FILTERS
        params = Hikari.GPUFilterParams(f)                                                                 # ref: Hikari.GPUFilterParams filter.jl:574
        data = params.filter_type <= Int32(2) ? nothing : Hikari.GPUFilterSamplerData(f)                   # ref: Hikari.GPUFilterSamplerData filter.jl:638
        out = Matrix{Float32}(undef, 15, length(px))                                                       # lambda4, pdf4, filter weight, o3, d3 (hk_test_camera's layout)
        for i in eachindex(px)
            x, y = px[i], py[i]
            ps = Hikari.compute_pixel_sample(rng, x, y, si[i])                                             # ref: Hikari.compute_pixel_sample sampler/sobol.jl:440
            fs = Hikari.filter_sample(params, data, Point2f(ps.jitter_x, ps.jitter_y))                     # ref: Hikari.filter_sample filter.jl:926
            lambda = Hikari.sample_wavelengths_visible(ps.wavelength_u)                                    # ref: Hikari.sample_wavelengths_visible spectral/spectral.jl:221
            p_film = Point2f(Float32(x) + 0.5f0 + fs.p[1], Float32(height) - Float32(y) + 1f0 + 0.5f0 + fs.p[2])
            ray, _ = Hikari.generate_ray(camera, Hikari.CameraSample(p_film, Point2f(ps.lens_u, ps.lens_v), ps.time))   # ref: Hikari.generate_ray camera/perspective.jl:95
            out[:, i] .= (lambda.lambda..., lambda.pdf..., fs.weight, ray.o[1], ray.o[2], ray.o[3], ray.d[1], ray.d[2], ray.d[3])
        end
        put!("camera_" * name, out)
    end
end

# ---- uplift --------------------------------------------------------------------------------------------------------------
function stage_uplift()
    rgb, lam = IN["uplift_rgb"], IN["uplift_lambda"]
    n = size(rgb, 2)
    o = [Matrix{Float32}(undef, 4, n) for _ in 1:3]
    for i in 1:n
        c = Hikari.RGBSpectrum(rgb[1, i], rgb[2, i], rgb[3, i])
        l = wl(lam, i)
        o[1][:, i] .= spec4(Hikari.uplift_rgb(TABLE, c, l))                                                # ref: Hikari.uplift_rgb spectral/uplift.jl:348
        o[2][:, i] .= spec4(Hikari.uplift_rgb_unbounded(TABLE, c, l))                                      # ref: Hikari.uplift_rgb_unbounded spectral/uplift.jl:368
        o[3][:, i] .= spec4(Hikari.uplift_rgb_illuminant(TABLE, c, l))                                     # ref: Hikari.uplift_rgb_illuminant spectral/uplift.jl:555
    end
    put!("uplift_bounded", o[1]); put!("uplift_unbounded", o[2]); put!("uplift_illuminant", o[3])
end

# ---- BSDFs: the palette of tests/test_reference_fixtures.py::reference_palette, index for index ----------------------------------
R(x...) = Hikari.RGBSpectrum(Float32.(x)...)
palette() = [
    Hikari.MatteMaterial(Kd=R(0.6, 0.4, 0.2)),                                                             # 0
    Hikari.MatteMaterial(Kd=R(0.6, 0.4, 0.2), σ=20f0),                                                     # 1
    Hikari.MirrorMaterial(Kr=R(0.9, 0.8, 0.7)),                                                            # 2
    Hikari.GlassMaterial(Kr=R(0.9), Kt=R(0.8, 0.9, 1.0), index=1.5f0),                                     # 3
    Hikari.ConductorMaterial(eta=R(0.2, 0.92, 1.1), k=R(3.9, 2.45, 2.14), roughness=0.09f0),               # 4 rough
    Hikari.ConductorMaterial(eta=R(0.2, 0.92, 1.1), k=R(3.9, 2.45, 2.14), roughness=0f0),                  # 5 smooth
    Hikari.Gold(roughness=0.04f0),                                                                         # 6 measured eta / k
    Hikari.CoatedDiffuseMaterial(reflectance=R(0.5, 0.3, 0.2), roughness=0.1f0, thickness=0.01f0, eta=1.5f0, albedo=R(0.0), g=0f0, max_depth=10, n_samples=1),      # 7
    Hikari.CoatedDiffuseMaterial(reflectance=R(0.4, 0.5, 0.6), roughness=0f0, thickness=0.05f0, eta=1.33f0, albedo=R(0.6, 0.7, 0.8), g=0.3f0, max_depth=10, n_samples=2),   # 8 smooth coat, scattering slab
    Hikari.ThinDielectricMaterial(eta=1.5f0),                                                              # 9
    Hikari.DiffuseTransmissionMaterial(reflectance=R(0.5, 0.4, 0.3), transmittance=R(0.3, 0.4, 0.5), scale=1f0),   # 10
    Hikari.CoatedDiffuseTransmissionMaterial(reflectance=R(0.5, 0.3, 0.2), transmittance=R(0.2, 0.3, 0.4), roughness=0.15f0, thickness=0.01f0, eta=1.5f0, albedo=R(0.0), g=0f0, max_depth=10, n_samples=1),   # 11
    Hikari.CoatedConductorMaterial(interface_roughness=0.05f0, interface_eta=1.5f0, conductor_eta=R(0.2, 0.92, 1.1), conductor_k=R(3.9, 2.45, 2.14), conductor_roughness=0.1f0,
                                   thickness=0.01f0, albedo=R(0.0), g=0f0, max_depth=10, n_samples=1),      # 12
    Hikari.CoatedConductorMaterial(interface_roughness=0f0, interface_eta=1.5f0, conductor_eta=R(0.2, 0.92, 1.1), conductor_k=R(3.9, 2.45, 2.14), conductor_roughness=0f0,
                                   thickness=0.01f0, albedo=R(0.0), g=0f0, max_depth=10, n_samples=1),      # 13 smooth / smooth (Q21)
]
function stage_bsdf()
    wo, wi, ns, lam, u, uc = IN["bsdf_wo"], IN["bsdf_wi"], IN["bsdf_ns"], IN["bsdf_lambda"], IN["bsdf_u"], IN["bsdf_uc"]
    textures = ()
    for (k, mat) in enumerate(palette()), regularize in (false, true)
        s = Matrix{Float32}(undef, 10, N)        # wi3, f4, pdf, is_specular, eta_scale (hk_test_bsdf mode 0)
        e = Matrix{Float32}(undef, 10, N)        # f4, pdf, 0 ... (mode 1)
        fill!(e, 0f0)
        for i in 1:N
            tfc = Hikari.TextureFilterContext(Point2f(0f0, 0f0))                                           # ref: Hikari.TextureFilterContext textures/texture-ref.jl:21
            l = wl(lam, i)
            b = Hikari.sample_bsdf_spectral(mat, TABLE, textures, v3(wo, i), v3(ns, i), tfc, l, p2(u, i), uc[i], regularize)   # ref: Hikari.sample_bsdf_spectral materials/spectral-eval.jl:42
            s[:, i] .= (b.wi[1], b.wi[2], b.wi[3], spec4(b.f)..., b.pdf, b.is_specular ? 1f0 : 0f0, b.eta_scale)
            if !regularize
                f, pdf = Hikari.evaluate_bsdf_spectral(mat, TABLE, textures, v3(wo, i), v3(wi, i), v3(ns, i), tfc, l)           # ref: Hikari.evaluate_bsdf_spectral materials/spectral-eval.jl:371
                e[1:5, i] .= (spec4(f)..., pdf)
            end
        end
        put!("bsdf_sample_$(k - 1)_reg$(Int(regularize))", s)
        regularize || put!("bsdf_eval_$(k - 1)", e)
    end
end

# ---- lights: the scene of tests/test_reference_fixtures.py::reference_light_scene -------------------------------------------------
function light_scene()
    scene = Hikari.Scene()
    white = Hikari.MatteMaterial(Kd=R(0.73, 0.73, 0.73))
    push!(scene, Hikari.AmbientLight(R(0.5, 0.6, 0.9)))                                                    # ref: Hikari.AmbientLight lights/ambient.jl:34
    push!(scene, Hikari.SpotLight(Point3f(-0.7f0, 1.7f0, -0.8f0), Point3f(0.1f0, 0.3f0, 0.1f0), R(20, 18, 14), 25f0, 15f0))   # ref: Hikari.SpotLight lights/spot.jl:49
    push!(scene, Hikari.DirectionalLight(R(2.0, 1.9, 1.6), Vec3f(0.25f0, -0.45f0, 1f0)))                   # ref: Hikari.DirectionalLight lights/directional.jl:68
    push!(scene, Hikari.PointLight(Point3f(0.5f0, 1.6f0, -0.4f0), R(6, 5, 3)))                             # ref: Hikari.PointLight lights/point.jl:26
    push!(scene, Hikari.SunLight(R(3, 2.8, 2.5), Vec3f(-0.3f0, -0.8f0, 0.2f0)))                            # ref: Hikari.SunLight lights/sun.jl:43
    push!(scene, normal_mesh(Rect3f(Vec3f(-1, 0, -1), Vec3f(2, 0.01f0, 2))), white)
    # emissive panels: 6 x 4 thin boxes under the ceiling, every face emissive -> 24 x 12 = 288 DiffuseAreaLights in the light BVH
    for ix in 0:5, iz in 0:3
        x0, z0 = -0.9f0 + 0.3f0 * ix, -0.6f0 + 0.3f0 * iz
        quad = Rect3f(Vec3f(x0, 1.97f0, z0), Vec3f(0.2f0, 0.005f0, 0.2f0))
        Le = R(0.2 + 0.1 * ix, 0.9 - 0.1 * iz, 0.5)
        push!(scene, normal_mesh(quad), Hikari.Emissive(Le=Le, scale=1f0 + 0.25f0 * iz, two_sided=false))  # ref: Hikari.Emissive materials/emissive.jl:55
    end
    Hikari.sync!(scene)
    scene
end
function stage_lights()
    scene = light_scene()
    lights = scene.lights
    sampler = Hikari.BVHLightSampler(lights; scene_radius=Hikari.world_radius(scene))                      # ref: Hikari.BVHLightSampler lights/bvh-light-sampler.jl:283
    nodes = sampler.nodes
    nn = length(nodes)
    nd = Matrix{Float32}(undef, 16, nn)          # hk_scene_light_bvh_copy's record: bmin3, bmax3, w3, phi, cos_o, cos_e, two_sided, child1_or_light, is_leaf, 0
    for (i, n) in enumerate(nodes)
        nd[:, i] .= (n.bounds_min..., n.bounds_max..., n.w..., n.phi, n.cosθ_o, n.cosθ_e, n.two_sided ? 1f0 : 0f0, Float32(n.child1_or_light_idx), n.is_leaf ? 1f0 : 0f0, 0f0)
    end
    put!("lightbvh_nodes", nd)
    put!("lightbvh_bit_trails", Vector{UInt32}(sampler.light_to_bit_trail))
    put!("lightbvh_counts", Int32[sampler.num_bvh_lights, sampler.num_infinite_lights, length(lights)])
    p, n, u1 = IN["light_p"], IN["light_n"], IN["light_u1"]
    chosen = Vector{Int32}(undef, N); pmf = Vector{Float32}(undef, N); qpmf = Vector{Float32}(undef, N)
    query = Int32[Int32(1 + (i * 7) % length(lights)) for i in 1:N]
    for i in 1:N
        chosen[i], pmf[i] = Hikari.bvh_sample_light(nodes, sampler.infinite_light_indices, sampler.num_infinite_lights, sampler.num_bvh_lights, p3(p, i), v3(n, i), u1[i])   # ref: Hikari.bvh_sample_light lights/bvh-light-sampler.jl:105
        qpmf[i] = Hikari.bvh_pmf(nodes, sampler.light_to_bit_trail, sampler.num_infinite_lights, sampler.num_bvh_lights, p3(p, i), v3(n, i), query[i])                       # ref: Hikari.bvh_pmf lights/bvh-light-sampler.jl:184
    end
    put!("lightbvh_choice", chosen); put!("lightbvh_pmf", pmf); put!("lightbvh_query", query); put!("lightbvh_query_pmf", qpmf)
    # sample_light_spectral of every light (flat index), one light per block of points
    static = Raycore.get_static(lights)
    lam, u2 = IN["light_lambda"], IN["light_u2"]
    ls = Matrix{Float32}(undef, 12, N)           # wi3, pdf, Li4, p_light3, is_delta (hko_light mode 0)
    which = Vector{Int32}(undef, N)
    for i in 1:N
        flat = Int32(1 + (i - 1) % length(lights))
        s = Hikari.sample_light_spectral(TABLE, static, flat, p3(p, i), wl(lam, i), p2(u2, i))             # ref: Hikari.sample_light_spectral physical-wavefront/lights.jl:386
        ls[:, i] .= (s.wi..., s.pdf, spec4(s.Li)..., s.p_light..., s.is_delta ? 1f0 : 0f0)
        which[i] = flat
    end
    put!("light_sample", ls); put!("light_index", which)
end

# ---- EnvironmentLight: Distribution2D sampling, equal-area mapping, nearest-texel radiance (environment_map.jl, sampling.jl:179-361) ------------
function stage_envlight()
    e = IN["env_rgb"]                                        # NumPy [v, u, rgb] -> Julia (3, u, v)
    h, w = size(e, 3), size(e, 2)
    data = [R(e[1, u, v], e[2, u, v], e[3, u, v]) for v in 1:h, u in 1:w]      # data[v, u]
    env = Hikari.EnvironmentMap(data)                                                                      # ref: Hikari.EnvironmentMap textures/environment_map.jl:9
    light = Hikari.EnvironmentLight(env, R(0.8, 1.0, 1.2))                                                 # ref: Hikari.EnvironmentLight lights/environment.jl:5
    p, lam, u2 = IN["light_p"], IN["light_lambda"], IN["light_u2"]
    out = Matrix{Float32}(undef, 12, N)                      # wi3, pdf, Li4, p_light3, is_delta
    for i in 1:N
        s = Hikari.sample_light_spectral(TABLE, (), light, p3(p, i), wl(lam, i), p2(u2, i))                # ref: Hikari.sample_light_spectral physical-wavefront/lights.jl:158
        out[:, i] .= (s.wi..., s.pdf, spec4(s.Li)..., s.p_light..., s.is_delta ? 1f0 : 0f0)
    end
    put!("envlight_sample", out)
    D = env.distribution
    put!("envlight_marginal_cdf", Vector{Float32}(Array(D.marginal_cdf)))
    put!("envlight_conditional_cdf", Vector{Float32}(vec(Array(D.conditional_cdf))))
end

# ---- NanoVDB -------------------------------------------------------------------------------------------------------------
function stage_nanovdb()
    d = permutedims(IN["nvdb_density"], (3, 2, 1))           # the input is [nx, ny, nz] in NumPy order
    bounds = Raycore.Bounds3(Point3f(-0.5f0, 0f0, -0.3f0), Point3f(0.5f0, 0.6f0, 0.2f0))
    m = Hikari.NanoVDBMedium(d; bounds=bounds, σ_a=R(0.0), σ_s=R(1.0), g=0.877f0, majorant_res=Hikari.Vec3i(16, 16, 16))   # ref: Hikari.NanoVDBMedium volpath/nanovdb.jl:964
    put!("nvdb_buffer", Vector{UInt8}(m.buffer))
    put!("nvdb_majorant", Vector{Float32}(vec(Array(m.majorant_grid.voxels))))
    ijk = IN["nvdb_ijk"]
    vals = Vector{Float32}(undef, size(ijk, 2))
    for i in eachindex(vals)
        vals[i] = Hikari.nanovdb_get_value(m, (), (ijk[1, i], ijk[2, i], ijk[3, i]))                       # ref: Hikari.nanovdb_get_value volpath/nanovdb.jl:315
    end
    put!("nvdb_values", vals)
    pw, lam = IN["nvdb_p"], IN["light_lambda"]
    sp = Matrix{Float32}(undef, 13, size(pw, 2))             # sigma_a4, sigma_s4, Le4, g (hk_test_medium mode 0)
    for i in axes(pw, 2)
        mp = Hikari.sample_point(m, (), TABLE, p3(pw, i), wl(lam, i))                                      # ref: Hikari.sample_point volpath/nanovdb.jl:477
        sp[:, i] .= (spec4(mp.σ_a)..., spec4(mp.σ_s)..., spec4(mp.Le)..., mp.g)
    end
    put!("nvdb_sample_point", sp)
    put!("nvdb_index_bbox", Int32[m.index_min..., m.index_max...])
end

# ---- the frame of test/volpath_integration.jl:30-90 ------------------------------------------------------------------------------
function integration_scene(; with_fog::Bool)
    white = Hikari.MatteMaterial(Kd=R(0.73, 0.73, 0.73)); red = Hikari.MatteMaterial(Kd=R(0.65, 0.05, 0.05)); green = Hikari.MatteMaterial(Kd=R(0.12, 0.45, 0.15))
    glass = Hikari.GlassMaterial(Kr=R(1.0), Kt=R(1.0), index=1.5f0)
    fog = Hikari.HomogeneousMedium(σ_a=R(0.01), σ_s=R(0.3), Le=R(0.0), g=0.3f0)
    gold = Hikari.ConductorMaterial(eta=R(0.15557, 0.42415, 1.3831), k=R(3.6024, 2.4721, 1.9155))
    box, half = 2f0, 1f0
    scene = Hikari.Scene()
    add!(prim, mat) = push!(scene, normal_mesh(prim isa Sphere ? Tesselation(prim, 32) : prim), mat)
    add!(Rect3f(Vec3f(-half, 0, -half), Vec3f(box, 0.01f0, box)), white)
    add!(Rect3f(Vec3f(-half, 0, half - 0.01f0), Vec3f(box, box, 0.01f0)), white)
    add!(Rect3f(Vec3f(-half, 0, -half), Vec3f(0.01f0, box, box)), red)
    add!(Rect3f(Vec3f(half - 0.01f0, 0, -half), Vec3f(0.01f0, box, box)), green)
    add!(Sphere(Point3f(-0.4f0, 0.4f0, 0f0), 0.35f0), with_fog ? Hikari.MediumInterface(glass; inside=fog, outside=nothing) : glass)
    add!(Sphere(Point3f(0.4f0, 0.35f0, 0f0), 0.3f0), gold)
    push!(scene, Hikari.PointLight(Point3f(0f0, 1.8f0, 0f0), R(15.0)))
    Hikari.sync!(scene)
    scene
end
function frame(scene, res, spp, depth)
    film = Hikari.Film(Point2f(res, res))
    camera = Hikari.PerspectiveCamera(Point3f(0f0, 1f0, -3.5f0), Point3f(0f0, 1f0, 0f0), film; fov=40f0)
    Hikari.clear!(film)
    vp = Hikari.VolPath(samples=spp, max_depth=depth)                                                      # ref: Hikari.VolPath volpath/volpath.jl:75
    vp(scene, film, camera)
    fb = Array(film.framebuffer)                              # Matrix{RGB{Float32}}[h, w], linear HDR (volpath.jl:415)
    out = Array{Float32}(undef, 3, size(fb, 2), size(fb, 1))  # -> NumPy [h, w, 3]
    for r in axes(fb, 1), c in axes(fb, 2)
        out[:, c, r] .= (fb[r, c].r, fb[r, c].g, fb[r, c].b)
    end
    out
end
function stage_frame()
    put!("frame_surfaces_64_spp4_depth5", frame(integration_scene(with_fog=false), 64, 4, 5))
    put!("frame_fog_32_spp1024_depth4", frame(integration_scene(with_fog=true), 32, 1024, 4))
end

# ---- run -------------------------------------------------------------------------------------------------------------------------
stages = isempty(ARGS) ? ["sobol", "camera", "uplift", "bsdf", "lights", "envlight", "nanovdb", "frame"] : ARGS
for s in stages
    @info "reference fixtures: $s"
    getfield(@__MODULE__, Symbol("stage_" * s))()
end
write_set(OUT_DIR, OUT)
mv(OUT_DIR * "_table.tmp", joinpath(OUT_DIR, "srgb_spectrum_table.dat"); force=true)
@info "wrote $(length(OUT)) arrays to $OUT_DIR — now run: python -m pytest tests/test_reference_fixtures.py"
