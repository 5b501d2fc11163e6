# CMBLensingHIPExt.jl -- glue that puts libcmblens_hip.so (MI355X / gfx950) behind CMBLensing.jl's operator surface.
#
# STATUS: written against CMBLensing.jl v0.10.1 by reading its sources; **never executed** -- no Julia runtime exists in the build
# image or on the GPU boxes.  What IS executed is the same set of C entry points through the Python mirror
# (cmblensing.jl_amd/, ctypes) and through the plain-C caller tests/c_abi/lenseflow.c.  Citations are file:line of the reference.
#
# What plugs in where
#   * `HIPLenseFlow <: FlowOpWithAdjoint` takes the `ds.L` operator slot (src/dataset.jl:55, `load_sim(L = HIPLenseFlow)`):
#     `L(ϕ)*f`, `L(ϕ)\f`, `L(ϕ)'*g`, `L(ϕ)'\g` (src/flowops.jl:11-14) and the two Zygote pullbacks (src/flowops.jl:40-68) are one
#     `ccall` each.  MAP_joint / MAP_marg / sample_joint / argmaxf_logpdf run unmodified on top (they only use that surface).
#   * `HIPDataSet` wraps a `BaseDataSet` and overrides the two documented performance hooks, `gradientf_logpdf`
#     (src/dataset.jl:76-80) and `argmaxf_logpdf` (src/maximization.jl:17-42, the Wiener-filter CG), with `cmbl_gradientf_logpdf`
#     / `cmbl_wiener_cg`; everything else is forwarded to the wrapped dataset.
# Fields cross the boundary as device pointers of `ROCArray`-backed `.arr` (AMDGPU.jl); layouts are the reference's own
# (Ny, Nx, Npol, Nbatch) column-major arrays (src/proj_cartesian.jl:13-36), so nothing is copied or permuted.
module CMBLensingHIPExt

using CMBLensing, AMDGPU, LinearAlgebra, Random, Zygote
using CMBLensing: FlowOpWithAdjoint, BaseDataSet, DataSet, Field, BaseField, ProjLambert, FuncOp, Map, Fourier, Ł, Ð,
                  LenseBasis, DerivBasis, batch_length, unbatch, nan2zero, diag
import CMBLensing: precompute!!, getϕ, gradientf_logpdf, argmaxf_logpdf
import Base: *, \, adjoint

const lib = get(ENV, "CMBL_LIB", joinpath(@__DIR__, "..", "cmblensing.jl_amd", "libcmblens_hip.so"))

# ---- status codes -> exceptions (include/cmblens.h: nothing throws across the ABI) ---------------------------------------
chk(rc::Integer) = rc == 0 ? nothing : error("libcmblens_hip error $rc: ", unsafe_string(ccall((:cmbl_last_error, lib), Cstring, ())))

const MAP, FOURIER, HARMONIC = Cint(0), Cint(1), Cint(2)                  # CMBL_MAP / CMBL_FOURIER / CMBL_HARMONIC
const FLOW_FWD, FLOW_INV, FLOW_ADJ, FLOW_INVADJ = Cint(0), Cint(1), Cint(2), Cint(3)
dtype(::Type{Float32}) = Cint(0)
dtype(::Type{Float64}) = Cint(1)
devptr(a::ROCArray) = Ptr{Cvoid}(UInt(pointer(a)))
npol(f::BaseField) = size(f.arr, 3)
nbatch(f::BaseField) = size(f.arr, 4)

# the library's basis tag of a field: Map-like, QU/IQU-Fourier ("FOURIER") or EB/IEB-Fourier ("HARMONIC")
basis_tag(::BaseField{B}) where {B<:CMBLensing.SpatialBasis{Map}} = MAP
basis_tag(::BaseField{B}) where {B<:Union{Fourier,CMBLensing.QUFourier,CMBLensing.IQUFourier}} = FOURIER
basis_tag(::BaseField{B}) where {B<:Union{CMBLensing.EBFourier,CMBLensing.IEBFourier}} = HARMONIC

# ---- context: replaces the memoized ProjLambert + FFT plans (src/proj_lambert.jl:48-75, src/util_fft.jl:32-39) ----------------
mutable struct HIPContext
    h :: Ptr{Cvoid}
    function HIPContext(proj::ProjLambert{T}) where {T}
        h = Ref{Ptr{Cvoid}}()
        chk(ccall((:cmbl_ctx_create, lib), Cint, (Cint, Cint, Cdouble, Cint, Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                  proj.Ny, proj.Nx, proj.θpix, dtype(real(T)), AMDGPU.device_id(AMDGPU.device()) - 1, AMDGPU.stream().stream, h))
        finalizer(c -> ccall((:cmbl_ctx_destroy, lib), Cint, (Ptr{Cvoid},), c.h), new(h[]))
    end
end
const contexts = IdDict{Any,HIPContext}()                                 # one per (memoized, hence ===) ProjLambert
hip_ctx(proj::ProjLambert) = get!(() -> HIPContext(proj), contexts, proj)

# ---- the LenseFlow operator ---------------------------------------------------------------------------------------------
# same abstract parent as LenseFlow (src/lenseflow.jl:2,19-31); `nsteps` RK4 steps, t: 0 -> 1
mutable struct HIPLenseFlow{T} <: FlowOpWithAdjoint{T}
    ϕ      :: Union{Nothing,Field}
    nsteps :: Int
    ctx    :: Union{Nothing,HIPContext}
    h      :: Ptr{Cvoid}
    cached :: Any                       # the ϕ object the device cache was built from (src/lenseflow.jl:123-129)
    alias_quirk :: Bool                 # true = the δϕ velocity exactly as written upstream (src/lenseflow.jl:198-200 aliasing)
end
HIPLenseFlow(nsteps::Int=7; alias_quirk=false) = ϕ -> HIPLenseFlow(ϕ, nsteps; alias_quirk)
function HIPLenseFlow(ϕ::Field, nsteps::Int=7; alias_quirk=false)
    T = real(eltype(ϕ))
    ctx = hip_ctx(ϕ.metadata)
    h = Ref{Ptr{Cvoid}}()
    chk(ccall((:cmbl_lenseflow_create, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), ctx.h, nsteps, h))
    L = HIPLenseFlow{T}(ϕ, nsteps, ctx, h[], nothing, alias_quirk)
    finalizer(L -> ccall((:cmbl_lenseflow_destroy, lib), Cint, (Ptr{Cvoid},), L.h), L)
end
getϕ(L::HIPLenseFlow) = L.ϕ
(L::HIPLenseFlow)(ϕ::Field) = (L.ϕ === ϕ) ? L : (L.ϕ = ϕ; L)              # `L(ϕ)`: re-points the operator, cache rebuilt lazily

# precompute!! (src/lenseflow.jl:80-142).  ϕ is handed over in the basis it arrives in -- `gradhess(ϕ)` differentiates a Fourier ϕ
# WITHOUT projecting it through a map first (src/specialops.jl:184-188, src/lenseflow.jl:135), and a ϕ produced by a gradient step
# has ky = 0 / Nyquist rows that no real map produces; `Map(ϕ)` here would change the next MAP step by 2e-5 (DESIGN.md §3).
function precompute!!(L::HIPLenseFlow, f)
    if L.cached !== L.ϕ
        ϕ = L.ϕ
        ϕ′ = (basis_tag(ϕ) == MAP) ? ϕ : Fourier(ϕ)
        chk(ccall((:cmbl_lenseflow_set_phi, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
                  L.h, basis_tag(ϕ′), devptr(ϕ′.arr), nbatch(ϕ′)))
        L.cached = ϕ
    end
    L
end

function flow(L::HIPLenseFlow, mode, f::BaseField, out::BaseField)
    precompute!!(L, f)
    chk(ccall((:cmbl_lenseflow_apply, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint),
              L.h, mode, basis_tag(f), devptr(f.arr), basis_tag(out), devptr(out.arr), npol(f), nbatch(f)))
    out
end
# src/flowops.jl:11-14: L*f, L\f act in the LenseBasis (maps), L'*g, L'\g in the DerivBasis (QU-Fourier); the library converts
*(L::HIPLenseFlow, f::Field) = (g = Ł(f); flow(L, FLOW_FWD, g, similar(g)))
\(L::HIPLenseFlow, f::Field) = (g = Ł(f); flow(L, FLOW_INV, g, similar(g)))
*(L::Adjoint{<:Any,<:HIPLenseFlow}, f::Field) = (g = Ð(f); flow(parent(L), FLOW_ADJ, g, similar(g)))
\(L::Adjoint{<:Any,<:HIPLenseFlow}, f::Field) = (g = Ð(f); flow(parent(L), FLOW_INVADJ, g, similar(g)))

# the δ-flow pullback: (δϕ [Fourier S0], δf [same basis as Δ], f_start [Map]) from the primal OUTPUT f_end and the cotangent Δ
function flow_gradient(L::HIPLenseFlow, mode, f_end::BaseField, Δ::BaseField)
    precompute!!(L, f_end)
    δf = similar(Δ)
    fstart = similar(f_end)
    ϕ = L.ϕ
    δϕ = similar(Fourier(ϕ), complex(real(eltype(ϕ))), (size(Fourier(ϕ).arr)[1:3]..., nbatch(f_end)))   # (Ny÷2+1, Nx, 1, Nbatch)
    δϕ = typeof(Fourier(ϕ))(δϕ, ϕ.metadata)
    chk(ccall((:cmbl_lenseflow_grad, lib), Cint,
              (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint),
              L.h, mode, devptr(f_end.arr), basis_tag(Δ), devptr(Δ.arr), devptr(δϕ.arr), basis_tag(δf), devptr(δf.arr),
              devptr(fstart.arr), npol(f_end), nbatch(f_end), L.alias_quirk ? 1 : 0))
    δϕ, δf, fstart
end

# the two Zygote adjoints of src/flowops.jl:40-68, including the :AD_constants shortcut (ϕ held constant -> plain adjoint flow)
Zygote.@adjoint function *(Lϕ::HIPLenseFlow, f::Field{B}) where {B}
    f̃ = Lϕ * f
    function back(Δ)
        if :ϕ in get(task_local_storage(), :AD_constants, ())
            nothing, B(Lϕ' * Δ)
        else
            δϕ, δf, _ = flow_gradient(Lϕ, FLOW_FWD, Ł(f̃), Ð(Δ))          # δ-flow t: 1 -> 0 from (f̃, Δ, 0)
            δϕ, B(δf)
        end
    end
    f̃, back
end
Zygote.@adjoint function \(Lϕ::HIPLenseFlow, f̃::Field{B}) where {B}
    f = Lϕ \ f̃
    function back(Δ)
        if :ϕ in get(task_local_storage(), :AD_constants, ())
            nothing, B(Lϕ' \ Δ)
        else
            δϕ, δf, _ = flow_gradient(Lϕ, FLOW_INV, Ł(f), Ð(Δ))           # δ-flow t: 0 -> 1 from (f, Δ, 0)
            δϕ, B(δf)
        end
    end
    f, back
end
# `L(ϕ)` inside a differentiated function: the cotangent of the operator is the cotangent of ϕ (src/flowops.jl:18-19)
Zygote.@adjoint (Lϕ::HIPLenseFlow)(ϕ′) = Lϕ(ϕ′), Δ -> (nothing, Δ)

# ---- data model + Wiener filter ---------------------------------------------------------------------------------------------
# include/cmblens.h: CMBL_OP_*
const OP_CF_INV, OP_CN_INV, OP_B, OP_MF, OP_D, OP_D_INV, OP_PRECOND_INV, OP_CPHI_INV, OP_G_INV, OP_MPIX = Cint.(0:9)

"""
    HIPDataSet(ds::BaseDataSet)

`ds` (at fiducial θ) with its Fourier-diagonal operators resident in the library: `gradientf_logpdf` and `argmaxf_logpdf`
(the Wiener filter) then run entirely inside libcmblens_hip (`cmbl_gradientf_logpdf`, `cmbl_wiener_cg`).  `ds.L` must be a
`HIPLenseFlow` (constructor or instance).  The diagonals are the `diag(...)` arrays of the reference operators in the harmonic
basis, `pinv` taken here exactly as `Hessian_logpdf_preconditioner(:f, ds)` does (src/dataset.jl:129-132).
"""
struct HIPDataSet{DS<:BaseDataSet} <: DataSet
    ds :: DS
    h  :: Ptr{Cvoid}
    L  :: HIPLenseFlow
end
Base.getproperty(d::HIPDataSet, k::Symbol) = k in (:ds, :h, :L) ? getfield(d, k) : getproperty(getfield(d, :ds), k)
(d::HIPDataSet)(θ) = isempty(θ) ? d : error("HIPDataSet holds the operators at fiducial θ; rebuild it from ds(θ)")

harmonic_planes(D::Diagonal) = devptr(real.(diag(CMBLensing.HarmonicBasis(D))).arr), size(diag(D).arr, 3)      # (Ny÷2+1, Nx, P) real planes
function set_op!(h, which, D)
    p, n = harmonic_planes(D)
    chk(ccall((:cmbl_dataset_set_op, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), h, which, p, n))
end
function HIPDataSet(ds::BaseDataSet)
    P = size(ds.d.arr, 3)
    ctx = hip_ctx(ds.d.metadata)
    h = Ref{Ptr{Cvoid}}()
    chk(ccall((:cmbl_dataset_create, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), ctx.h, P, h))
    set_op!(h[], OP_CF_INV, pinv(ds.Cf));  set_op!(h[], OP_CN_INV, pinv(ds.Cn))
    set_op!(h[], OP_B, ds.B);              set_op!(h[], OP_D, ds.D);   set_op!(h[], OP_D_INV, pinv(ds.D))
    set_op!(h[], OP_PRECOND_INV, pinv(pinv(ds.Cf) + ds.B̂' * ds.M̂' * pinv(ds.Cn̂) * ds.M̂ * ds.B̂))
    set_op!(h[], OP_CPHI_INV, pinv(ds.Cϕ)); set_op!(h[], OP_G_INV, pinv(ds.G))
    # M = Mfourier * Mpix (src/dataset.jl:279-285): the Fourier part is diagonal, the pixel mask is a (Ny, Nx) map
    Mf, Mpix = ds.M.a, ds.M.b
    set_op!(h[], OP_MF, Mf)
    chk(ccall((:cmbl_dataset_set_op, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), h[], OP_MPIX, devptr(diag(Mpix).arr), 1))
    d = CMBLensing.HarmonicBasis(ds.d)
    chk(ccall((:cmbl_dataset_set_data, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), h[], devptr(d.arr), nbatch(d)))
    chk(ccall((:cmbl_dataset_set_logdet, lib), Cint, (Ptr{Cvoid}, Cdouble), h[], logdet(ds.Cf) + logdet(ds.Cϕ) + logdet(ds.Cn)))
    L = ds.L isa HIPLenseFlow ? ds.L : HIPLenseFlow(zero(diag(ds.Cϕ)), 7)
    HIPDataSet(ds, h[], L)
end

# src/dataset.jl:76-80:  L'B'M'Cn⁻¹(d − M B L f) − Cf⁻¹ f, one library call
function gradientf_logpdf(hd::HIPDataSet; f, ϕ, θ=(;), d=hd.ds.d)
    L = precompute!!(hd.L(ϕ), f)
    fh, dh = CMBLensing.HarmonicBasis(f), CMBLensing.HarmonicBasis(d)
    out = similar(fh)
    zero_d = all(iszero, dh.arr) ? 1 : 0
    chk(ccall((:cmbl_gradientf_logpdf, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
              hd.h, L.h, devptr(fh.arr), devptr(dh.arr), zero_d, devptr(out.arr), nbatch(fh)))
    out
end

# src/maximization.jl:17-42: the preconditioned CG of src/numerical_algorithms.jl:73-134 with its scalars on the device;
# returns (f, history) like the reference (`history_keys = (:i, :res)`)
function argmaxf_logpdf(hd::HIPDataSet, Ω::NamedTuple, d=hd.ds.d; fstart=nothing, preconditioner=:diag,
                        conjgrad_kwargs=(tol=1e-1, nsteps=500), offset=false)
    offset && error("offset=true is the reference's generic path: a₀ ≡ 0 for this linear model")
    L = precompute!!(hd.L(Ω.ϕ), d)
    dh = CMBLensing.HarmonicBasis(d)
    out = similar(dh)
    nsteps = get(conjgrad_kwargs, :nsteps, 500)
    hist = Vector{Cdouble}(undef, nsteps * nbatch(dh))
    nit = Ref{Cint}(0)
    fs = isnothing(fstart) ? C_NULL : devptr(CMBLensing.HarmonicBasis(fstart).arr)
    chk(ccall((:cmbl_wiener_cg, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cint}, Cint),
              hd.h, L.h, devptr(dh.arr), fs, get(conjgrad_kwargs, :tol, 1e-1), nsteps, devptr(out.arr), hist, nit, nbatch(dh)))
    B = nbatch(dh)
    history = [(i=i, res=(B == 1 ? hist[i] : CMBLensing.batch(hist[(i-1)*B+1:i*B]))) for i in 1:nit[]]
    out, history
end

# ---- reductions and random fields (optional: the generic Julia broadcasts on ROCArrays work too) ------------------------------
function LinearAlgebra.dot(a::BaseField{B}, b::BaseField{B}) where {B}
    ctx = hip_ctx(a.metadata)
    out = Vector{Cdouble}(undef, nbatch(a))
    chk(ccall((:cmbl_dot, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}),
              ctx.h, basis_tag(a), devptr(a.arr), devptr(b.arr), npol(a), nbatch(a), out))
    nbatch(a) == 1 ? out[1] : CMBLensing.batch(out)
end

# `set_sum_accuracy_mode!` (src/util.jl:288-292) for the library's reductions: nothing / Float64 / :kahan
function set_sum_accuracy_mode!(proj::ProjLambert, mode)
    m = mode === nothing ? 0 : mode === Float64 ? 1 : mode === :kahan ? 2 : error("mode must be `nothing`, `:kahan`, `Float64`")
    chk(ccall((:cmbl_set_sum_accuracy_mode, lib), Cint, (Ptr{Cvoid}, Cint), hip_ctx(proj).h, m))
end

# device RNG for `simulate` / `randn!` (src/specialops.jl:6, src/base_fields.jl:169-170): counter-based Philox4x32-10
mutable struct HIPPhilox <: Random.AbstractRNG
    seed   :: UInt64
    stream :: UInt64
end
function Random.randn!(rng::HIPPhilox, ξ::BaseField{B}) where {B<:CMBLensing.SpatialBasis{Map}}
    seeds = fill(rng.seed, 1)
    chk(ccall((:cmbl_randn, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Cint, UInt64, Ptr{Cvoid}, Clong),
              hip_ctx(ξ.metadata).h, seeds, 1, rng.stream, devptr(ξ.arr), length(ξ.arr)))
    rng.stream += 1
    ξ
end

end # module
