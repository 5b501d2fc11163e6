# test_hipext.jl -- the first thing to run on a machine that has Julia, CMBLensing.jl v0.10.1, AMDGPU.jl and an MI355X:
# every binding of CMBLensingHIPExt.jl against the reference's own CPU path on the same inputs.
#
#   julia --project=<env with CMBLensing, AMDGPU, Zygote, Adapt> julia/test_hipext.jl
#
# STATUS: **never executed** (no Julia runtime in the build image or on the GPU boxes); written by reading the reference's
# test/runtests.jl "Lensing" (:533-581) and "Posterior" (:585-621) sections, whose structure and tolerances it follows.  The C entry
# points it reaches are exercised through Python (tests/test_gpu_*.py) and plain C (tests/c_abi/*.c).
using Test, Random, LinearAlgebra
using CMBLensing, AMDGPU, Zygote
include(joinpath(@__DIR__, "CMBLensingHIPExt.jl"))
using .CMBLensingHIPExt: HIPLenseFlow, HIPDataSet, CMBL_ABI_VERSION

rel(a, b) = norm(Array(a.arr) .- Array(b.arr)) / norm(Array(b.arr))

@testset "libcmblens_hip through CMBLensingHIPExt" begin

    @test ccall((:cmbl_abi_version, CMBLensingHIPExt.lib), Cint, ()) == CMBL_ABI_VERSION

    for T in (Float32, Float64), pol in (:I, :P, :IP)
        flowtol, gradtol = T == Float32 ? (3e-5, 2e-4) : (1e-10, 1e-9)
        @testset "T=$T pol=$pol" begin
            # the reference's own simulated data set (src/dataset.jl:186-338), CPU storage
            (; f, ϕ, ds) = load_sim(; θpix=2, Nside=128, T, pol, seed=0, pixel_mask_kwargs=(; edge_padding_deg=0.4, apodization_deg=0.4, num_ptsrcs=0))
            Lcpu = LenseFlow(ϕ, 7)
            # the same fields in device storage (ext/CMBLensingCUDAExt.jl:42-43's twin `gpu`)
            fg, ϕg = gpu(f), gpu(ϕ)
            L = HIPLenseFlow(ϕg, 7)                                     # the DEFAULT is the reference's arithmetic (src/lenseflow.jl:198-200)
            @test L.alias_quirk === true && CMBLensingHIPExt.reference_exact()
            @test HIPLenseFlow(ϕg, 7; alias_quirk=false).alias_quirk === false    # the consistent form stays a keyword

            # operator surface (src/flowops.jl:11-14)
            @test rel(L * Map(fg), Lcpu * Map(f)) < flowtol
            @test rel(L \ Map(fg), Lcpu \ Map(f)) < flowtol
            @test rel(L' * Fourier(fg), Lcpu' * Fourier(f)) < flowtol
            @test rel(L' \ Fourier(fg), Lcpu' \ Fourier(f)) < flowtol
            # adjoint identity, as test/runtests.jl:556-562
            g = simulate(ds.Cf; seed=1); gg = gpu(g)
            @test dot(gg, L * fg) ≈ dot(fg, L' * gg) rtol = (T == Float32 ? 2e-4 : 1e-10)

            # Zygote pullbacks (src/flowops.jl:40-68) against the reference's own
            ∇cpu = gradient((f, ϕ) -> norm(LenseFlow(ϕ, 7) * f), Map(f), ϕ)
            ∇hip = gradient((f, ϕ) -> norm(HIPLenseFlow(ϕ, 7) * f), Map(fg), ϕg)          # default arguments: must reproduce the reference's gradient
            @test rel(∇hip[1], ∇cpu[1]) < flowtol
            @test rel(∇hip[2], ∇cpu[2]) < gradtol

            # data model: gradientf_logpdf, Wiener filter, logpdf(Mixed) and its gradient (src/dataset.jl:76-117, src/maximization.jl:17-42)
            hd = HIPDataSet(gpu(ds))
            @test rel(gradientf_logpdf(hd; f=fg, ϕ=ϕg), gradientf_logpdf(ds; f, ϕ)) < 4flowtol
            fwf_cpu, hist_cpu = argmaxf_logpdf(ds, (; ϕ); conjgrad_kwargs=(; tol=0, nsteps=8, history_keys=(:i, :res)))
            fwf_hip, hist_hip = argmaxf_logpdf(hd, (; ϕ=ϕg); conjgrad_kwargs=(; tol=0, nsteps=8, history_keys=(:i, :res)))
            @test rel(fwf_hip, fwf_cpu) < (T == Float32 ? 3e-4 : 1e-9)
            @test all(isapprox.(getindex.(hist_hip, :res), getindex.(hist_cpu, :res); rtol=(T == Float32 ? 2e-3 : 1e-8)))
            (f°, ϕ°) = mix(ds; f, ϕ)
            f°g, ϕ°g = gpu(f°), gpu(ϕ°)
            @test logpdf(Mixed(hd); f°=f°g, ϕ°=ϕ°g) ≈ logpdf(Mixed(ds); f°, ϕ°) rtol = (T == Float32 ? 2e-5 : 1e-10)
            gcpu = gradient((f°, ϕ°) -> logpdf(Mixed(ds); f°, ϕ°), f°, ϕ°)
            ghip = gradient((f°, ϕ°) -> logpdf(Mixed(hd); f°, ϕ°), f°g, ϕ°g)
            @test rel(ghip[1], gcpu[1]) < gradtol
            @test rel(ghip[2], gcpu[2]) < 3gradtol

            # the drivers run unmodified on top (src/maximization.jl:116-233): one MAP_joint step each side
            m_cpu = MAP_joint(ds; nsteps=1, progress=false)
            m_hip = MAP_joint(hd; nsteps=1, progress=false)
            @test rel(m_hip.ϕ, m_cpu.ϕ) < (T == Float32 ? 2e-2 : 1e-6)
        end
    end
end
