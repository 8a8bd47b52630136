"""hikari.jl_amd — MI355X-native VolPath hot path of JuliaGraphics/Hikari.jl behind a C-ABI.

The package holds only what the path needs: `csrc/` (HIP kernels + the C-ABI library) and this host-side
mirror of the reference's Integrator / Scene / Material / Film interface for the path.  The directory name
contains a dot, so import it through the `hikari_jl_amd` shim at the repository root.
"""
from . import _abi, geometry, tables
from ._lib import HikariMI355XError, LIB_PATH
from .camera import MatrixCamera, PerspectiveCamera
from .envmap import Distribution2D, EnvironmentLight, EnvironmentMap, analytic_sky, rotation_matrix
from .denoise import DenoiseConfig
from .film import Film
from .lights import (AmbientLight, DiffuseAreaLight, DirectionalLight, PointLight, RGBIlluminantSpectrum, SpotLight,
                     SunLight)
from .materials import (Aluminum, Brass, Copper, Gold, Silver, CoatedConductorMaterial, CoatedDiffuseMaterial, CoatedDiffuseTransmissionMaterial,
                        ConductorMaterial, DiffuseTransmissionMaterial, Emissive, GlassMaterial, MatteMaterial,
                        MediumInterface, MirrorMaterial, MixMaterial, PiecewiseLinearSpectrum, PlasticMaterial,
                        RGBSpectrum, Texture, ThinDielectricMaterial, VertexColorTexture)
from .media import GridMedium, HomogeneousMedium, NanoVDBMedium, RGBGridMedium
from .media_presets import Coffee, Fog, Juice, Milk, Smoke, SubsurfaceMedium, Wine, get_medium_preset
from .postprocess import FilmSensor, compute_white_balance_matrix
from .scene import Scene
from .sunsky import sunsky_to_envlight
from .volpath import (BoxFilter, Comm, Context, GaussianFilter, LanczosSincFilter, MitchellFilter, TriangleFilter, VolPath,
                      integrator_params, scene_handle)
