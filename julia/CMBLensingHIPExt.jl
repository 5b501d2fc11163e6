# CMBLensingHIPExt.jl -- glue that puts libcmblens_hip.so (MI355X / gfx950) behind CMBLensing.jl's operator surface.
#
# STATUS: written against CMBLensing.jl v0.10.1 by reading its sources; **never executed** -- no Julia runtime exists in the build
# image or on the GPU boxes.  What IS executed is the same set of C entry points through the Python mirror
# (cmblensing.jl_amd/, ctypes) and through the plain-C callers tests/c_abi/*.c.  Citations are file:line of the reference.
# julia/test_hipext.jl (also unexecuted) is the first thing to run on a machine that has Julia + AMDGPU.jl (it compares every binding below with the
# reference's own CPU path); julia/make_reference_fixtures.jl needs no GPU at all.
#
# What plugs in where
#   1. storage level, the twin of ext/CMBLensingCUDAExt.jl:42-93 for `ROCArray`: `gpu`, `is_gpu_backed`, `Cℓ_to_2D`, `pinv` / `inv`
#      of Diagonals, `fill!`, `sum`, CPU-RNG `randn!` into device memory, `unsafe_free!`.  With these alone the
#      reference runs on the GPU through AMDGPU.jl broadcasts + rocFFT plans (AbstractFFTs dispatches on the array type,
#      src/util_fft.jl:32-35); everything below replaces the hot path on top.
#   2. `HIPLenseFlow <: FlowOpWithAdjoint` takes the `ds.L` operator slot (src/dataset.jl:55, `load_sim(L = HIPLenseFlow)`):
#      `L(ϕ)*f`, `L(ϕ)\f`, `L(ϕ)'*g`, `L(ϕ)'\g` (src/flowops.jl:11-14) and the two Zygote pullbacks (src/flowops.jl:40-68) are one
#      `ccall` each.  MAP_joint / MAP_marg / sample_joint / argmaxf_logpdf run unmodified on top (they only use that surface).
#   3. `HIPDataSet` wraps a `BaseDataSet` and overrides the performance hooks `gradientf_logpdf` (src/dataset.jl:76-80) and
#      `argmaxf_logpdf` (src/maximization.jl:17-42, the Wiener-filter CG) with `cmbl_gradientf_logpdf` / `cmbl_wiener_cg`, and the
#      mixed posterior `logpdf(Mixed(ds); f°, ϕ°)` with its gradient (src/dataset.jl:84-87; called by MAP_joint at
#      src/maximization.jl:178,197,205 and by hmc_step through src/sampling.jl:399) with `cmbl_logpdf_mixed` /
#      `cmbl_grad_logpdf_mixed` -- the call bench.py times.  Everything else is forwarded to the wrapped dataset.
# Fields cross the boundary as device pointers of `ROCArray`-backed `.arr` (AMDGPU.jl); layouts are the reference's own
# (Ny, Nx, Npol, Nbatch) column-major arrays (src/proj_cartesian.jl:13-36), so nothing is copied or permuted.  Every array whose
# pointer is passed is rooted with `GC.@preserve` for the duration of the call (calls are stream-ordered: a temporary that is
# only an INPUT of an asynchronous call is additionally kept until `cmbl_ctx_synchronize`, see `keepalive`).
module CMBLensingHIPExt

using CMBLensing, AMDGPU, Adapt, LinearAlgebra, Random, Zygote
using CMBLensing: FlowOpWithAdjoint, BaseDataSet, DataSet, Mixed, Field, BaseField, ProjLambert, Map, Fourier, EBFourier, IEBFourier,
                  QUFourier, IQUFourier, Ł, Ð, BlockDiagIEB, LazyBinaryOp, FieldTuple, batch_length, batch, unbatch, nan2zero, diag
import CMBLensing: precompute!!, getϕ, gradientf_logpdf, argmaxf_logpdf, logpdf
import Base: *, \, adjoint

const lib = get(ENV, "CMBL_LIB", joinpath(@__DIR__, "..", "cmblensing.jl_amd", "libcmblens_hip.so"))

# DEFAULT ARITHMETIC OF THIS GLUE = THE REFERENCE'S, AS WRITTEN: the δϕ velocity with the in-place aliasing of src/lenseflow.jl:198-200 and
# plain sums in the working precision (`sum_accuracy_mode = nothing`, src/util.jl:288-316) -- a user who swaps `LenseFlow` for
# `HIPLenseFlow` gets the reference's numbers, not the library's "consistent" variant (DESIGN.md §3 Q1: that one matches finite differences
# to 2e-8 and differs from the reference by ~3e-4 in the ϕ gradient).  The consistent form stays a keyword (`alias_quirk=false`), Float64 /
# Kahan accumulation a call (`set_sum_accuracy_mode!`); CMBL_CONSISTENT=1 makes both the default of a session.
reference_exact() = get(ENV, "CMBL_CONSISTENT", "0") in ("", "0")

# ---- status codes -> exceptions (include/cmblens.h: nothing throws across the ABI) ---------------------------------------
chk(rc::Integer) = rc == 0 ? nothing : error("libcmblens_hip error $rc: ", unsafe_string(ccall((:cmbl_last_error, lib), Cstring, ())))
const CMBL_ABI_VERSION = 3          # include/cmblens.h: the revision these ccall signatures were written against
function __init__()
    # CMBL_REFERENCE_EXACT (the switch of the Python host and of earlier revisions of this glue) stays an accepted alias: =0 means CMBL_CONSISTENT=1
    if haskey(ENV, "CMBL_REFERENCE_EXACT") && !haskey(ENV, "CMBL_CONSISTENT")
        ENV["CMBL_CONSISTENT"] = ENV["CMBL_REFERENCE_EXACT"] in ("", "0") ? "1" : "0"
    end
    v = ccall((:cmbl_abi_version, lib), Cint, ())
    v == CMBL_ABI_VERSION || error("libcmblens_hip.so has ABI version $v, this extension binds version $CMBL_ABI_VERSION: rebuild one of them")
end

const MAP, FOURIER, HARMONIC = Cint(0), Cint(1), Cint(2)                  # CMBL_MAP / CMBL_FOURIER / CMBL_HARMONIC
const FLOW_FWD, FLOW_INV, FLOW_ADJ, FLOW_INVADJ = Cint(0), Cint(1), Cint(2), Cint(3)
dtype(::Type{Float32}) = Cint(0)
dtype(::Type{Float64}) = Cint(1)

const ROCBaseField{B,M,T,A<:ROCArray} = BaseField{B,M,T,A}
devptr(a::ROCArray) = Ptr{Cvoid}(UInt(pointer(a)))
npol(f::BaseField) = size(f.arr, 3)
nbatch(f::BaseField) = size(f.arr, 4)

# the library's basis tag of a field: Map-like, QU/IQU-Fourier ("FOURIER") or EB/IEB-Fourier ("HARMONIC"); spin-0 Fourier is both
basis_tag(::BaseField{B}) where {B<:CMBLensing.SpatialBasis{Map}} = MAP
basis_tag(::BaseField{B}) where {B<:Union{Fourier,QUFourier,IQUFourier}} = FOURIER
basis_tag(::BaseField{B}) where {B<:Union{EBFourier,IEBFourier}} = HARMONIC
# the basis covariances are diagonal in.  NB `CMBLensing.HarmonicBasis(f)` keeps the pol basis (QU stays QU, src/generic.jl:94-98),
# so the conversion is spelled out here
harm(f::BaseField) = npol(f) == 1 ? Fourier(f) : npol(f) == 2 ? EBFourier(f) : IEBFourier(f)

# ---- 1. storage-level twins of ext/CMBLensingCUDAExt.jl:42-93 --------------------------------------------------------------
CMBLensing.is_gpu_backed(::ROCBaseField) = true                                                   # :42
CMBLensing.gpu(x) = Adapt.adapt_structure(ROCArray, x)                                            # :43
function CMBLensing.Cℓ_to_2D(Cℓ, proj::ProjLambert{T,<:ROCArray}) where {T}                       # :46-49 (through the CPU, like upstream)
    CMBLensing.gpu(T.(nan2zero.(Cℓ.(CMBLensing.cpu(proj.ℓmag)))))
end
LinearAlgebra.pinv(D::Diagonal{T,<:ROCBaseField}) where {T} = Diagonal(@. ifelse(isfinite(inv(D.diag)), inv(D.diag), $zero(T)))   # :55
LinearAlgebra.inv(D::Diagonal{T,<:ROCBaseField}) where {T} =
    any(Array((D.diag .== 0)[:])) ? throw(SingularException(-1)) : Diagonal(inv.(D.diag))        # :56
Base.fill!(f::ROCBaseField, x) = (fill!(f.arr, x); f)                                             # :57
Base.sum(f::ROCBaseField; dims=:) =
    (dims == :) ? CMBLensing.sum_dropdims(f.arr) : (1 in dims) ? error("Sum over invalid dims of a flat field.") : f      # :58
Random.randn!(rng::MersenneTwister, A::ROCArray) = (A .= adapt(ROCArray, randn!(rng, adapt(Array, A))))   # :72-73 host RNG + upload (upstream's own "minor type-piracy", kept: `simulate` with the CPU generator needs it)
CMBLensing.unsafe_free!(x::ROCArray) = AMDGPU.unsafe_free!(x)                                     # :88
# (:91-93, `dot(x::CuArray, y::CuArray) = sum(conj.(x) .* y)`, works around a CUDA.jl / Zygote issue and has no twin here: AMDGPU.jl's own
# `dot` is left alone -- redefining it for two ROCArrays would be type piracy on a package this module does not own)

# ---- context: replaces the memoized ProjLambert + FFT plans (src/proj_lambert.jl:48-75, src/util_fft.jl:32-39) ----------------
mutable struct HIPContext
    h    :: Ptr{Cvoid}
    keep :: Vector{Any}              # inputs of calls that are still in flight on the stream (see `keepalive`)
    function HIPContext(proj::ProjLambert{T}) where {T}
        h = Ref{Ptr{Cvoid}}()
        # AMDGPU.jl: `AMDGPU.stream()` is the task-local HIPStream, `.stream` its hipStream_t; `AMDGPU.device_id` is 1-based
        chk(ccall((:cmbl_ctx_create, lib), Cint, (Cint, Cint, Cdouble, Cint, Cint, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                  proj.Ny, proj.Nx, proj.θpix, dtype(real(T)), AMDGPU.device_id(AMDGPU.device()) - 1,
                  Ptr{Cvoid}(UInt(Base.unsafe_convert(Ptr{Cvoid}, AMDGPU.stream().stream))), h))
        # the library's own default accumulates in Float64; the reference's is the working precision (src/util.jl:288-316)
        reference_exact() && chk(ccall((:cmbl_set_sum_accuracy_mode, lib), Cint, (Ptr{Cvoid}, Cint), h[], 0))
        finalizer(c -> ccall((:cmbl_ctx_destroy, lib), Cint, (Ptr{Cvoid},), c.h), new(h[], Any[]))
    end
end
const contexts = IdDict{Any,HIPContext}()                                 # one per (memoized, hence ===) ProjLambert
hip_ctx(proj::ProjLambert) = get!(() -> HIPContext(proj), contexts, proj)
hip_ctx(f::BaseField) = hip_ctx(f.metadata)
synchronize(ctx::HIPContext) = (chk(ccall((:cmbl_ctx_synchronize, lib), Cint, (Ptr{Cvoid},), ctx.h)); empty!(ctx.keep); nothing)
# a temporary that is only read by an asynchronous call must outlive the call, not just the `ccall`: park it on the context; the
# list is dropped at the next synchronisation (every entry point that returns host values synchronises)
keepalive(ctx::HIPContext, xs...) = (append!(ctx.keep, xs); length(ctx.keep) > 256 && synchronize(ctx); nothing)

# ---- 2. the LenseFlow operator -------------------------------------------------------------------------------------------
# same abstract parent as LenseFlow (src/lenseflow.jl:2,19-31); `nsteps` RK4 steps, t: 0 -> 1
mutable struct HIPLenseFlow{T} <: FlowOpWithAdjoint{T}
    ϕ      :: Union{Nothing,Field}
    nsteps :: Int
    ctx    :: Union{Nothing,HIPContext}
    h      :: Ptr{Cvoid}
    cached :: Any                       # the ϕ object the device cache was built from (src/lenseflow.jl:123-129)
    alias_quirk :: Bool                 # true = the δϕ velocity exactly as written upstream (src/lenseflow.jl:198-200 aliasing)
end
HIPLenseFlow(nsteps::Int=7; alias_quirk=reference_exact()) = ϕ -> HIPLenseFlow(ϕ, nsteps; alias_quirk)
function HIPLenseFlow(ϕ::Field, nsteps::Int=7; alias_quirk=reference_exact())
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
        a = ϕ′.arr
        GC.@preserve a chk(ccall((:cmbl_lenseflow_set_phi, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
                                 L.h, basis_tag(ϕ′), devptr(a), nbatch(ϕ′)))
        keepalive(L.ctx, a)
        L.cached = ϕ
    end
    L
end

function flow(L::HIPLenseFlow, mode, f::BaseField, out::BaseField)
    precompute!!(L, f)
    a, o = f.arr, out.arr
    GC.@preserve a o chk(ccall((:cmbl_lenseflow_apply, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint),
                               L.h, mode, basis_tag(f), devptr(a), basis_tag(out), devptr(o), npol(f), nbatch(f)))
    keepalive(L.ctx, a)
    out
end
# src/flowops.jl:11-14: L*f, L\f act in the LenseBasis (maps), L'*g, L'\g in the DerivBasis (QU-Fourier); the library converts
*(L::HIPLenseFlow, f::Field) = (g = Ł(f); flow(L, FLOW_FWD, g, similar(g)))
\(L::HIPLenseFlow, f::Field) = (g = Ł(f); flow(L, FLOW_INV, g, similar(g)))
*(L::Adjoint{<:Any,<:HIPLenseFlow}, f::Field) = (g = Ð(f); flow(parent(L), FLOW_ADJ, g, similar(g)))
\(L::Adjoint{<:Any,<:HIPLenseFlow}, f::Field) = (g = Ð(f); flow(parent(L), FLOW_INVADJ, g, similar(g)))

# an uninitialised Fourier spin-0 field with the batch length of `like` (for δϕ: (Ny÷2+1, Nx, 1, Nbatch))
function similar_ϕ(ϕ::Field, like::BaseField)
    ϕf = Fourier(ϕ)
    typeof(ϕf)(similar(ϕf.arr, eltype(ϕf.arr), (size(ϕf.arr, 1), size(ϕf.arr, 2), 1, nbatch(like))), ϕf.metadata)
end

# the δ-flow pullback: (δϕ [Fourier S0], δf [same basis as Δ], f_start [Map]) from the primal OUTPUT f_end and the cotangent Δ
function flow_gradient(L::HIPLenseFlow, mode, f_end::BaseField, Δ::BaseField)
    precompute!!(L, f_end)
    δf, fstart, δϕ = similar(Δ), similar(f_end), similar_ϕ(L.ϕ, f_end)
    a, b, c, d, e = f_end.arr, Δ.arr, δϕ.arr, δf.arr, fstart.arr
    GC.@preserve a b c d e chk(ccall((:cmbl_lenseflow_grad, lib), Cint,
              (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Cint),
              L.h, mode, devptr(a), basis_tag(Δ), devptr(b), devptr(c), basis_tag(δf), devptr(d),
              devptr(e), npol(f_end), nbatch(f_end), L.alias_quirk ? 1 : 0))
    keepalive(L.ctx, a, b)
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

# ---- 3. data model, Wiener filter, mixed posterior ------------------------------------------------------------------------
# include/cmblens.h: CMBL_OP_*
const OP_CF_INV, OP_CN_INV, OP_B, OP_MF, OP_D, OP_D_INV, OP_PRECOND_INV, OP_CPHI_INV, OP_G_INV, OP_MPIX = Cint.(0:9)

"""
    HIPDataSet(ds::BaseDataSet)

`ds` evaluated at its current θ with the Fourier-diagonal operators resident in the library: `gradientf_logpdf`, `argmaxf_logpdf`
(the Wiener filter), `logpdf(Mixed(ds); f°, ϕ°)` and its gradient then run entirely inside libcmblens_hip.  `ds.L` must be a
`HIPLenseFlow` (constructor or instance).  The planes handed over are the `diag(...)` arrays of the reference operators in the
harmonic (E/B) basis -- five planes (TT, TE, ET, EE, BB) for a `BlockDiagIEB` -- with `pinv` taken here exactly as
`Hessian_logpdf_preconditioner(:f, ds)` does (src/dataset.jl:129-132).  `hd(θ)` / `copy(hd)` / `hd.G = I` work like on any
`DataSet` (src/dataset.jl:5,12-18; MAP_joint does all three, src/maximization.jl:145-146) and re-upload what changed.
"""
mutable struct HIPDataSet{DS<:BaseDataSet} <: DataSet
    ds :: DS
    h  :: Ptr{Cvoid}
    L  :: HIPLenseFlow
end
Base.getproperty(d::HIPDataSet, k::Symbol) = k in (:ds, :h, :L) ? getfield(d, k) : getproperty(getfield(d, :ds), k)
function Base.setproperty!(d::HIPDataSet, k::Symbol, v)
    k in (:ds, :h, :L) && return setfield!(d, k, v)
    setproperty!(getfield(d, :ds), k, v)
    upload!(d)                                                            # e.g. `dsθ.G = I` (src/maximization.jl:146)
    v
end
Base.copy(d::HIPDataSet) = HIPDataSet(copy(getfield(d, :ds)))
# ds(θ) (src/dataset.jl:12-18): only when some operator of the wrapped dataset actually depends on a key of θ are the operators
# re-evaluated and uploaded (MAP_joint passes θ to every logpdf call of an already evaluated dsθ, src/maximization.jl:178,197,205)
depends_on_θ(d::HIPDataSet, θ) = !isempty(θ) && any(v -> CMBLensing.depends_on(v, θ), CMBLensing.fieldvalues(getfield(d, :ds)))
(d::HIPDataSet)(θ::NamedTuple) = depends_on_θ(d, θ) ? HIPDataSet(getfield(d, :ds)(θ)) : d
(d::HIPDataSet)(; θ...) = d((; θ...))

# real planes of an operator that is diagonal in the harmonic basis, as ONE (Ny÷2+1, Nx, nplanes) device array
op_planes(D::Diagonal{<:Any,<:BaseField}) = real.(harm(D.diag).arr[:, :, :, 1])
op_planes(D::BlockDiagIEB) = cat((real.(diag(X).arr[:, :, 1, 1]) for X in (D.ΣTE[1,1], D.ΣTE[1,2], D.ΣTE[2,1], D.ΣTE[2,2], D.ΣB))...; dims=3)
op_planes(::UniformScaling, like) = fill!(similar(op_planes(like)), 1)    # `G = I`
function set_op!(hd_h, ctx, which, D, like=nothing)
    p = D isa UniformScaling ? op_planes(D, like) : op_planes(D)
    GC.@preserve p chk(ccall((:cmbl_dataset_set_op, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), hd_h, which, devptr(p), size(p, 3)))
    keepalive(ctx, p)                                                     # the library copies the planes on its stream
end
function upload!(hd::HIPDataSet)
    ds, h = getfield(hd, :ds), getfield(hd, :h)
    ctx = hip_ctx(ds.d.metadata)
    Cf, Cn, Cϕ, D, G = ds.Cf, ds.Cn, ds.Cϕ, ds.D, ds.G                    # already evaluated at θ by ds(θ)
    set_op!(h, ctx, OP_CF_INV, pinv(Cf));  set_op!(h, ctx, OP_CN_INV, pinv(Cn))
    set_op!(h, ctx, OP_B, ds.B);           set_op!(h, ctx, OP_D, D);   set_op!(h, ctx, OP_D_INV, pinv(D))
    set_op!(h, ctx, OP_PRECOND_INV, pinv(pinv(Cf) + ds.B̂' * ds.M̂' * pinv(ds.Cn̂) * ds.M̂ * ds.B̂))
    set_op!(h, ctx, OP_CPHI_INV, pinv(Cϕ)); set_op!(h, ctx, OP_G_INV, G isa UniformScaling ? G : pinv(G), Cϕ)
    # M = Mfourier * Mpix (src/dataset.jl:279-285) is a LazyBinaryOp{*}(X = Mfourier, Y = Mpix) (src/specialops.jl:364-377) or,
    # without a pixel mask, the Fourier-diagonal operator alone
    M = ds.M
    if M isa LazyBinaryOp
        set_op!(h, ctx, OP_MF, M.X)
        m = Map(diag(M.Y)).arr[:, :, 1, 1]                                # the same mask on every pol plane (src/dataset.jl:281)
        GC.@preserve m chk(ccall((:cmbl_dataset_set_op, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), h, OP_MPIX, devptr(m), 1))
        keepalive(ctx, m)
    else
        set_op!(h, ctx, OP_MF, M)
    end
    d = harm(ds.d)
    a = d.arr
    GC.@preserve a chk(ccall((:cmbl_dataset_set_data, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), h, devptr(a), nbatch(d)))
    keepalive(ctx, a)
    chk(ccall((:cmbl_dataset_set_logdet, lib), Cint, (Ptr{Cvoid}, Cdouble), h, logdet(Cf) + logdet(Cϕ) + logdet(Cn)))
    hd
end
function HIPDataSet(ds::BaseDataSet)
    ctx = hip_ctx(ds.d.metadata)
    h = Ref{Ptr{Cvoid}}()
    chk(ccall((:cmbl_dataset_create, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cvoid}}), ctx.h, size(ds.d.arr, 3), h))
    L = ds.L isa HIPLenseFlow ? ds.L : HIPLenseFlow(zero(diag(ds.Cϕ)), 7)
    hd = HIPDataSet(ds, h[], L)
    finalizer(x -> ccall((:cmbl_dataset_destroy, lib), Cint, (Ptr{Cvoid},), getfield(x, :h)), hd)
    upload!(hd)
end

# src/dataset.jl:76-80:  L'B'M'Cn⁻¹(d − M B L f) − Cf⁻¹ f, one library call
function gradientf_logpdf(hd::HIPDataSet; f, ϕ, θ=(;), d=hd.ds.d)
    depends_on_θ(hd, θ) && return gradientf_logpdf(hd(θ); f, ϕ, d)
    L = precompute!!(hd.L(ϕ), f)
    fh, dh = harm(f), harm(d)
    out = similar(fh)
    zero_d = all(iszero, dh.arr) ? 1 : 0
    a, b, o = fh.arr, dh.arr, out.arr
    GC.@preserve a b o chk(ccall((:cmbl_gradientf_logpdf, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
                                 hd.h, L.h, devptr(a), devptr(b), zero_d, devptr(o), nbatch(fh)))
    keepalive(L.ctx, a, b)
    out
end

# src/maximization.jl:17-42: the preconditioned CG of src/numerical_algorithms.jl:73-134 with its scalars on the device;
# returns (f, history) like the reference (`history_keys = (:i, :res)`).  `offset=true` (used by sample_f, src/maximization.jl:56-62)
# adds a₀ = gradientf_logpdf(f = 0, d = 0) to b, which is identically 0 for this linear model.
function argmaxf_logpdf(hd::HIPDataSet, Ω::NamedTuple, d=hd.ds.d; fstart=nothing, preconditioner=:diag,
                        conjgrad_kwargs=(tol=1e-1, nsteps=500), offset=false)
    θ = get(Ω, :θ, (;))
    depends_on_θ(hd, θ) && return argmaxf_logpdf(hd(θ), Base.structdiff(Ω, NamedTuple{(:θ,)}), d; fstart, preconditioner, conjgrad_kwargs, offset)
    L = precompute!!(hd.L(Ω.ϕ), d)
    dh = harm(d)
    out = similar(dh)
    nsteps = get(conjgrad_kwargs, :nsteps, 500)
    B = nbatch(dh)
    hist = Vector{Cdouble}(undef, nsteps * B)
    nit = Ref{Cint}(0)
    fs = isnothing(fstart) ? nothing : harm(fstart).arr
    a, o = dh.arr, out.arr
    GC.@preserve a o fs hist chk(ccall((:cmbl_wiener_cg, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cint, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cint}, Cint),
              hd.h, L.h, devptr(a), isnothing(fs) ? C_NULL : devptr(fs), get(conjgrad_kwargs, :tol, 1e-1), nsteps, devptr(o), hist, nit, B))
    history = [(i=i, res=(B == 1 ? hist[i] : batch(hist[(i-1)*B+1:i*B]))) for i in 1:nit[]]
    out, history
end

# logpdf(Mixed(ds); f°, ϕ°[, θ]) (src/dataset.jl:84-87) and its gradient: unmix (G \ ϕ°, precompute, D \ (L \ f°)), the three
# quadratic forms with their logdets, and for the gradient the chain rule through one inverse and one forward δ-flow -- one call
# each.  The positional helper carries the Zygote adjoint (an `@adjoint` cannot return cotangents of keyword arguments; Zygote
# differentiates the keyword method below down to this call on its own), so `gradient(Ω° -> logpdf(Mixed(ds); f°, Ω°..., θ), Ω°)`
# (src/maximization.jl:178) and `gradient(U, ϕ°)` in hmc_step (src/sampling.jl:405) land on cmbl_grad_logpdf_mixed unmodified.
function hip_logpdf_mixed(hd::HIPDataSet, f°::Field, ϕ°::Field)
    fo, po = Ł(f°), Fourier(ϕ°)
    B = nbatch(fo)
    lp = Vector{Cdouble}(undef, B)
    a, b = fo.arr, po.arr
    GC.@preserve a b lp chk(ccall((:cmbl_logpdf_mixed, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint),
                                  hd.h, hd.L.h, devptr(a), devptr(b), lp, B))
    hd.L.cached = nothing                                                 # the library re-pointed the flow at G \ ϕ° (include/cmblens.h)
    T = real(eltype(fo))
    B == 1 ? T(lp[1]) : batch(T.(lp))
end
function hip_grad_logpdf_mixed(hd::HIPDataSet, f°::Field, ϕ°::Field)
    fo, po = Ł(f°), Fourier(ϕ°)
    B = nbatch(fo)
    lp = Vector{Cdouble}(undef, B)
    gf, gϕ = similar(fo), similar_ϕ(po, fo)
    a, b, c, d = fo.arr, po.arr, gf.arr, gϕ.arr
    GC.@preserve a b c d lp chk(ccall((:cmbl_grad_logpdf_mixed, lib), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint),
              hd.h, hd.L.h, devptr(a), devptr(b), lp, devptr(c), devptr(d), B, hd.L.alias_quirk ? 1 : 0))
    hd.L.cached = nothing
    T = real(eltype(fo))
    (B == 1 ? T(lp[1]) : batch(T.(lp))), gf, gϕ
end
Zygote.@adjoint function hip_logpdf_mixed(hd::HIPDataSet, f°::Field{Bf}, ϕ°::Field{Bϕ}) where {Bf,Bϕ}
    lp, gf, gϕ = hip_grad_logpdf_mixed(hd, f°, ϕ°)
    # gradients come back as the reference's do: the f° cotangent in the basis of f°, the ϕ° one in the basis of ϕ° (src/autodiff.jl:105-133)
    lp, Δ -> (nothing, Bf(Δ * gf), Bϕ(Δ * gϕ))
end
function logpdf(mds::Mixed{<:HIPDataSet}; f°, ϕ°, θ=(;), Ω...)
    lp = hip_logpdf_mixed(mds.ds(θ), f°, ϕ°)
    depends_on_θ(mds.ds, θ) ? lp - logdet(mds.ds.ds.D, θ) - logdet(mds.ds.ds.G, θ) : lp      # src/dataset.jl:86, on the θ-dependent originals
end

# ---- reductions and random fields (optional: the generic Julia broadcasts on ROCArrays work too) ------------------------------
# restricted to device-backed flat-sky fields: CPU fields and other projections keep the reference's own `dot`
function LinearAlgebra.dot(a::BaseField{B,<:ProjLambert,<:Any,<:ROCArray}, b::BaseField{B,<:ProjLambert,<:Any,<:ROCArray}) where {B}
    ctx = hip_ctx(a.metadata)
    out = Vector{Cdouble}(undef, nbatch(a))
    x, y = a.arr, b.arr
    GC.@preserve x y out chk(ccall((:cmbl_dot, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}),
                                   ctx.h, basis_tag(a), devptr(x), devptr(y), npol(a), nbatch(a), out))
    nbatch(a) == 1 ? out[1] : batch(out)
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
function Random.randn!(rng::HIPPhilox, ξ::BaseField{B,<:ProjLambert,<:Any,<:ROCArray}) where {B<:CMBLensing.SpatialBasis{Map}}
    seeds = fill(rng.seed, 1)
    a = ξ.arr
    GC.@preserve a seeds chk(ccall((:cmbl_randn, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Cint, UInt64, Ptr{Cvoid}, Clong),
                                   hip_ctx(ξ.metadata).h, seeds, 1, rng.stream, devptr(a), length(a)))
    rng.stream += 1
    ξ
end

end # module
