# make_reference_fixtures.jl -- run the REFERENCE (marius311/CMBLensing.jl v0.10.1, CPU, no GPU needed) on the committed inputs of
# tests/golden/ref_inputs/*.npy (tools/make_reference_inputs.py) and write its outputs to tests/golden/ref_outputs/*.npy.
#
#     julia --project=<an environment that has CMBLensing v0.10.1 and Zygote> julia/make_reference_fixtures.jl [repo root]
#
# STATUS: never executed (no Julia in the build image).  With the outputs present, tests/test_reference_fixtures.py compares the
# oracle (CPU) and the HIP engine (GPU, CMBL_REFERENCE_EXACT=1) with them and the parity of this repository is pinned by the
# reference itself instead of by the oracle alone; without them those tests skip with "parity unpinned".  Commit the outputs
# (≈3 MB): they are data.
#
# Files are plain NumPy .npy (version 1.0, little endian, C order); a Julia array (Ny, Nx, P, B) is written with the NumPy shape
# (B, P, Nx, Ny) -- the same bytes -- so nothing is permuted on either side.  No package beyond CMBLensing / Zygote is needed.
#
# What is computed (each the reference's own call; file:line of the reference in brackets)
#   flow_*       LenseFlow(ϕ,7)*f, \f [src/flowops.jl:11,13], '*g, '\g [:12,14], pullback of (ϕ,f) -> LenseFlow(ϕ,7)*f at cotangent g
#                [src/flowops.jl:40-53 -> src/lenseflow.jl:176-214], and norm-gradient gradient(ϕ -> norm(LenseFlow(ϕ,7)*f), ϕ)
#                [test/runtests.jl:533-581 use the same construct]
#   post_<pol>_* a dataset built by `load_sim` [src/dataset.jl:186-338] with OUR mask and data put in: logpdf(ds; f, ϕ)
#                [src/dataset.jl:59-66], mix [:96-101], logpdf(Mixed(ds); f°, ϕ°) and its gradient [:84-87, src/maximization.jl:178],
#                gradientf_logpdf [:76-80], an 8-iteration argmaxf_logpdf history [src/maximization.jl:17-42], the QE noise Nϕ that
#                load_sim stores [src/dataset.jl:312] and quadratic_estimate(ds) itself [src/quadratic_estimate.jl:29-52],
#                diag(D), one Hessian-preconditioned MAP_joint step [src/maximization.jl:160-206]
using CMBLensing, Zygote, LinearAlgebra, Random
using CMBLensing: BaseField, QUMap, IQUMap, EBFourier, IEBFourier, QUFourier, IQUFourier, Cℓ_to_Cov, LowPass, FieldTuple

root = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..")
indir, outdir = joinpath(root, "tests", "golden", "ref_inputs"), joinpath(root, "tests", "golden", "ref_outputs")
mkpath(outdir)

# ---- minimal .npy reader / writer ----------------------------------------------------------------------------------------
const NPY_T = Dict("<f8" => Float64, "<f4" => Float32, "<c16" => ComplexF64, "<c8" => ComplexF32, "<i8" => Int64)
function npy_read(path)
    open(path) do io
        @assert read(io, 6) == UInt8[0x93, 0x4e, 0x55, 0x4d, 0x50, 0x59] "not an .npy file: $path"
        major = read(io, UInt8); read(io, UInt8)
        hlen = major == 1 ? Int(read(io, UInt16)) : Int(read(io, UInt32))
        hdr = String(read(io, hlen))
        T = NPY_T[match(r"'descr':\s*'([^']+)'", hdr)[1]]
        @assert !occursin(r"'fortran_order':\s*True", hdr)
        shp = [parse(Int, m.match) for m in eachmatch(r"\d+", match(r"'shape':\s*\(([^)]*)\)", hdr)[1])]
        read!(io, Array{T}(undef, reverse(shp)...))                       # C order (B,P,Nx,Ny) == column-major (Ny,Nx,P,B)
    end
end
function npy_write(name, a::AbstractArray{T}) where {T}
    a = Array(a)
    descr = Dict(v => k for (k, v) in NPY_T)[T]
    shape = join(reverse(size(a)), ", ") * (ndims(a) == 1 ? "," : "")
    hdr = "{'descr': '$descr', 'fortran_order': False, 'shape': ($shape), }"
    hdr *= " "^(63 - (10 + length(hdr)) % 64) * "\n"                      # 10 = magic + version + header length field
    open(joinpath(outdir, name * ".npy"), "w") do io
        write(io, UInt8[0x93, 0x4e, 0x55, 0x4d, 0x50, 0x59, 0x01, 0x00], UInt16(length(hdr)), hdr, a)
    end
end
npy_write(name, x::Number) = npy_write(name, [x])
inp(name) = npy_read(joinpath(indir, name * ".npy"))
arr4(f::BaseField) = reshape(f.arr, size(f.arr, 1), size(f.arr, 2), size(f.arr, 3), size(f.arr, 4))
out(name, f::BaseField) = npy_write(name, arr4(f))
drop(a) = dropdims(a, dims=4)                                             # unbatched fields hold (Ny, Nx[, P]) arrays

# ---- flows -----------------------------------------------------------------------------------------------------------------
let
    ϕa, fa, ga = inp("flow_phi"), inp("flow_f"), inp("flow_g")           # (64,128,1,1), (64,128,2,1) ×2
    proj = ProjLambert(; Ny=size(ϕa, 1), Nx=size(ϕa, 2), θpix=2, T=Float64)
    ϕ = BaseField{Map}(ϕa[:, :, 1, 1], proj)
    f, g = BaseField{QUMap}(drop(fa), proj), BaseField{QUMap}(drop(ga), proj)
    L = LenseFlow(ϕ, 7)
    f̃ = L * f
    out("flow_Lf", QUMap(f̃))
    out("flow_Linvf", QUMap(L \ f))
    gl = QUFourier(g)
    out("flow_Ladjg", QUFourier(L' * gl))
    out("flow_Linvadjg", QUFourier(L' \ gl))
    _, back = Zygote.pullback((ϕ, f) -> LenseFlow(ϕ, 7) * f, ϕ, f)
    δϕ, δf = back(gl)
    out("flow_grad_dphi", Fourier(δϕ))
    out("flow_grad_df", QUFourier(δf))
    out("flow_gradnorm_dphi", Fourier(gradient(ϕ -> norm(LenseFlow(ϕ, 7) * f), ϕ)[1]))
    npy_write("flow_adjoint_identity", [dot(f, L * g), dot(L' * QUFourier(f), gl)])
end

# ---- posteriors ------------------------------------------------------------------------------------------------------------
for (pol, F, F̂) in ((:P, QUMap, EBFourier), (:IP, IQUMap, IEBFourier))
    tag = "post_$(pol)_"
    mask, fa, ϕa, da = inp(tag * "mask"), inp(tag * "f"), inp(tag * "phi"), inp(tag * "d")
    Ny, Nx, nF = size(mask, 1), size(mask, 2), size(fa, 3)
    T = Float64
    proj = ProjLambert(; Ny, Nx, θpix=3, T)
    ks = pol == :P ? (:EE, :BB) : (:TT, :EE, :BB, :TE)
    # M exactly as load_sim builds it (src/dataset.jl:276-291), with the committed mask in place of make_mask(...)
    Mfourier = Cℓ_to_Cov(pol, proj, ((k == :TE ? 0 : 1) * LowPass(3000).diag.Wℓ for k in ks)...; units=1)
    Mpix = Diagonal(BaseField{F}(cat(fill(mask, nF)...; dims=3), proj))
    sim = load_sim(; θpix=3, Nside=(Ny, Nx), pol, T, beamFWHM=3, M=Mfourier * Mpix, M̂=Mfourier, seed=0)
    ds = sim.ds
    f = BaseField{F̂}(drop(fa), proj)
    ϕ = BaseField{Fourier}(ϕa[:, :, 1, 1], proj)
    ds.d = BaseField{F̂}(drop(da), proj)
    out(tag * "Nphi", diag(ds.Nϕ))                                        # quadratic_estimate(ds).Nϕ / 2 of the SIMULATED data: data independent
    out(tag * "D", diag(ds.D()))
    npy_write(tag * "logpdf", [logpdf(ds; f, ϕ)])
    Ω° = mix(ds; f, ϕ)
    f°, ϕ° = Ω°.f°, Ω°.ϕ°
    out(tag * "fo", F(f°)); out(tag * "phio", Fourier(ϕ°))
    npy_write(tag * "logpdf_mixed", [logpdf(Mixed(ds); f°, ϕ°)])
    g = gradient((f°, ϕ°) -> logpdf(Mixed(ds); f°, ϕ°), f°, ϕ°)
    out(tag * "grad_fo", F(g[1])); out(tag * "grad_phio", Fourier(g[2]))
    out(tag * "gradientf", F̂(gradientf_logpdf(ds; f, ϕ)))
    fwf, hist = argmaxf_logpdf(ds, (; ϕ); conjgrad_kwargs=(tol=0, nsteps=8, history_keys=(:i, :res)))
    out(tag * "cg_f", F̂(fwf)); npy_write(tag * "cg_res", Float64[h.res for h in hist])
    qe = quadratic_estimate(ds)
    out(tag * "qe_phi", Fourier(qe.ϕqe)); out(tag * "qe_Nphi", diag(qe.Nϕ))
    # one MAP_joint step from ϕ = 0 (history keeps f, ϕ, α, logpdf; Brent's iterate sequence is Optim.jl's)
    m = MAP_joint(ds; nsteps=1, progress=false, history_keys=(:f, :ϕ, :α, :logpdf, :argmaxf_logpdf_history))
    h = m.history[end]
    out(tag * "map1_f", F̂(h.f)); out(tag * "map1_phi", Fourier(h.ϕ))
    npy_write(tag * "map1_alpha_logpdf_ncg", Float64[h.α, h.logpdf, length(h.argmaxf_logpdf_history)])
end
# ---- a chain file as JLD2.jl itself writes it (src/sampling.jl:311-320): `rundat` + `chunks_1 :: Vector{Vector{Any}}` of `Dict{Symbol,Any}` states ----
# tests/test_jld2_writer.py compares the file cmblensing.jl_amd/jld2_writer.py produces for the SAME content with this one, message by message
# (ADVICE r04: that writer's `Dict{Symbol,Any}` layout was inferred by analogy and has never been opened by JLD2.jl).  Plain values only (no Field
# structs), the keys and the step numbering of the reference's state (`:step` = 2 for the first Gibbs pass, `:logpdf`, `:ΔH`, `:accept`, `:ϕ`, `:f`, `:θ`).
using CMBLensing: jldopen
let fn = joinpath(outdir, "chain_fixture.jld2")
    state(step, c) = Dict{Symbol,Any}(:step => step, :logpdf => -100.0 - step - c, :ΔH => 0.25 * step, :accept => isodd(step), :ncg => 17.0,
                                      :ϕ => ComplexF64[x + 10y + im * c for x in 1:3, y in 1:4], :f => ComplexF64[x - y + im * step for x in 1:3, y in 1:4, p in 1:2],
                                      :θ => Dict{Symbol,Any}(:r => 0.21, :Aphi => 1.1))
    chunks = [Any[state(step, c) for step in 2:3] for c in 0:1]
    jldopen(fn, "w") do io
        write(io, "rundat", Dict{Symbol,Any}(:nchains => 2, :eps => 0.01, :rng => "device"))
        write(io, "chunks_1", chunks)
    end
    jldopen(fn, "a+") do io                                                # the append path (:313-316)
        write(io, "chunks_2", [Any[state(4, c)] for c in 0:1])
    end
end
println("wrote ", length(readdir(outdir)), " files to ", outdir)
