# MMCHip.jl -- the reference's hot-path methods served by libmmc_hip.so (MI355X / gfx950).
#
# Usage in Ewald/main.jl: after the `include("energy.jl")` / `include("ewalds.jl")` lines add
#
#     include("/path/to/metropolismontecarlo_amd/julia/MMCHip.jl")
#     using .MMCHip
#     MMCHip.attach!(moa, soa, vdwTable, box)      # once, after MakeAtomArrays / MakeTables
#
# `attach!` uploads the system; the methods below then shadow the reference's
#   LJ_poly_ΔU(i, moa, soa, vdwTable, r_cut, box)            Ewald/energy.jl:209-290
#   EwaldReal(chosenOne, moa, soa, ewald, r_cut, box)        Ewald/ewalds.jl:293-376
#   EwaldShort(i, moa, soa, sim_props, ewald, box)           Ewald/ewalds.jl:892-910
#   PrepareEwaldVariables(ewald, boxSize)                    Ewald/ewalds.jl:45-103
#   RecipLong(ewald, r, qq_q, box)                           Ewald/ewalds.jl:538-604
#   RecipMove(box, ewalds, r_old, r_new, qq_q)               Ewald/ewalds.jl:718-826
#   EwaldSelf(ewald, qq_q)                                   Ewald/ewalds.jl:829-833
#   potential(moa, soa, tot, ewalds, vdwTable, sim_props[, "ewald"])  Ewald/energy.jl:864-1032
# with identical argument lists and return values.  Host arrays are borrowed for the duration of
# each ccall (GC.@preserve); nothing is cached by pointer.
#
# This file could not be executed in the build image (no `julia` there); it is the binding a
# maintainer adds, kept next to the C header it binds.  The same C ABI is exercised by the Python
# mirror (metropolismontecarlo_amd/api.py), which the GPU tests drive.
module MMCHip

using StaticArrays

const libmmc = get(ENV, "MMC_HIP_LIB", joinpath(@__DIR__, "..", "libmmc_hip.so"))

struct MMCTotals
    energy::Float64; virial::Float64; coulomb::Float64
    lj::Float64; real::Float64; recip::Float64; self::Float64
    n_overlap::Int32; _pad::Int32
end

mutable struct Session
    ctx::Ptr{Cvoid}
    box::Float64
    last_mol::Int64
    ewald_key::Tuple
    s_old::Vector{ComplexF64}   # host copies of what the device S buffers hold
    s_new::Vector{ComplexF64}
end

const SESSION = Ref{Union{Nothing,Session}}(nothing)

function check(status::Int32)
    status == 0 && return
    msg = unsafe_string(ccall((:mmc_last_error, libmmc), Cstring, ()))
    status == 2 && throw(AssertionError(msg))      # a reference @assert
    error("libmmc_hip: status $status: $msg")
end

"Upload moa/soa/vdwTable once (mmc_upload_system)."
function attach!(moa, soa, vdwTable, box::Float64; device::Integer = 0)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mmc_ctx_create, libmmc), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                device, C_NULL, ctx))
    com, fa, la = moa.COM, moa.firstAtom, moa.lastAtom
    coords, atype, charge = soa.coords, soa.atype, soa.charge
    eps, sig = vdwTable.ϵᵢⱼ, vdwTable.σᵢⱼ
    GC.@preserve com fa la coords atype charge eps sig begin
        check(ccall((:mmc_upload_system, libmmc), Int32,
                    (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64},
                     Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Float64),
                    ctx[], length(com), length(coords), pointer(com), pointer(fa), pointer(la),
                    pointer(coords), pointer(atype), pointer(charge), size(eps, 1), pointer(eps),
                    pointer(sig), box))
    end
    SESSION[] = Session(ctx[], box, 0, (), ComplexF64[], ComplexF64[])
    return SESSION[]
end

function detach!()
    s = SESSION[]
    s === nothing && return
    ccall((:mmc_ctx_destroy, libmmc), Int32, (Ptr{Cvoid},), s.ctx)
    SESSION[] = nothing
end

session() = (s = SESSION[]; s === nothing ? error("MMCHip.attach!(moa, soa, vdwTable, box) first") : s)

# Loop() changes moa.COM[i] / soa.coords[first:last] of ONE molecule between calls and may have
# restored the previous one (main.jl:527,552,623-624): re-send both.
function sync_molecule!(s::Session, moa, soa, i::Int64)
    for m in unique((i, s.last_mol))
        m == 0 && continue
        com = moa.COM; coords = soa.coords
        f = moa.firstAtom[m]
        GC.@preserve com coords begin
            check(ccall((:mmc_set_molecule, libmmc), Int32,
                        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}),
                        s.ctx, m, pointer(com, m), pointer(coords, f)))
        end
    end
    s.last_mol = i
end

function sync_all!(s::Session, moa, soa)
    com = moa.COM; coords = soa.coords
    GC.@preserve com coords begin
        check(ccall((:mmc_update_system, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                    s.ctx, pointer(com), pointer(coords)))
    end
    s.last_mol = 0
end

function bind_ewald!(s::Session, ewald, box)
    key = (ewald.kappa, ewald.nk, ewald.k_sq_max, ewald.factor, box)
    if s.ewald_key != key
        n = Ref{Int64}(0)
        check(ccall((:mmc_prepare_ewald, libmmc), Int32,
                    (Ptr{Cvoid}, Float64, Int64, Int64, Float64, Float64, Ptr{Int64}),
                    s.ctx, ewald.kappa, ewald.nk, ewald.k_sq_max, box, ewald.factor, n))
        s.ewald_key = key
        s.s_old = ComplexF64[]; s.s_new = ComplexF64[]   # device arrays were zeroed
    end
end

# Loop rebinds ewald.sumQExpOld/New to fresh copies (main.jl:621,628): push them when the
# arrays are not the ones the device mirrors.
function push_s!(s::Session, ewald)
    so, sn = ewald.sumQExpOld, ewald.sumQExpNew
    # compare CONTENT (337 complex numbers): array identity can be recycled by the allocator
    if so != s.s_old || sn != s.s_new
        GC.@preserve so sn begin
            check(ccall((:mmc_set_sumqexp, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                        s.ctx, pointer(so), pointer(sn)))
        end
        s.s_old = copy(so); s.s_new = copy(sn)
    end
end

function pull_s!(s::Session, ewald; old::Bool = false)
    so, sn = ewald.sumQExpOld, ewald.sumQExpNew
    GC.@preserve so sn begin
        check(ccall((:mmc_get_sumqexp, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                    s.ctx, old ? pointer(so) : C_NULL, pointer(sn)))
    end
    if old
        s.s_old = copy(so)
    end
    s.s_new = copy(sn)
end

# ---- the reference's methods -------------------------------------------------------------------

function PrepareEwaldVariables(ewald, boxSize::Real)
    box = Float64(min(boxSize...))
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mmc_ctx_create, libmmc), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), 0, C_NULL, ctx))
    n = Ref{Int64}(0)
    st = ccall((:mmc_prepare_ewald, libmmc), Int32,
               (Ptr{Cvoid}, Float64, Int64, Int64, Float64, Float64, Ptr{Int64}),
               ctx[], ewald.kappa, ewald.nk, ewald.k_sq_max, box, ewald.factor, n)
    st != 0 && (ccall((:mmc_ctx_destroy, libmmc), Int32, (Ptr{Cvoid},), ctx[]); check(st))
    kxyz = Vector{SVector{3,Int32}}(undef, n[])
    cfac = Vector{Float64}(undef, n[])
    GC.@preserve kxyz cfac begin
        check(ccall((:mmc_get_kvectors, libmmc), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}),
                    ctx[], pointer(kxyz), pointer(cfac)))
    end
    ccall((:mmc_ctx_destroy, libmmc), Int32, (Ptr{Cvoid},), ctx[])
    return typeof(ewald)(ewald.kappa, ewald.nk, ewald.k_sq_max, n[], kxyz, cfac,
                         zeros(ComplexF64, n[]), zeros(ComplexF64, n[]), ewald.factor)
end

function LJ_poly_ΔU(i, moa, soa, vdwTable, r_cut, box)
    s = session(); sync_molecule!(s, moa, soa, Int64(i))
    pot = Ref{Float64}(0.0); vir = Ref{Float64}(0.0)
    check(ccall((:mmc_lj_poly_du, libmmc), Int32,
                (Ptr{Cvoid}, Int64, Float64, Ptr{Float64}, Ptr{Float64}),
                s.ctx, i, r_cut, pot, vir))
    return pot[], vir[]
end

function EwaldReal(chosenOne::Int64, moa, soa, ewald, r_cut::Float64, box::Float64)
    s = session(); bind_ewald!(s, ewald, box); sync_molecule!(s, moa, soa, chosenOne)
    pot = Ref{Float64}(0.0); ov = Ref{Int32}(0)
    check(ccall((:mmc_ewald_real, libmmc), Int32,
                (Ptr{Cvoid}, Int64, Float64, Float64, Ptr{Float64}, Ptr{Int32}),
                s.ctx, chosenOne, r_cut, 0.5, pot, ov))
    return pot[], ov[] != 0
end

function EwaldShort(i::Int64, moa, soa, sim_props, ewald, box::Float64)
    s = session(); bind_ewald!(s, ewald, box); sync_molecule!(s, moa, soa, i)
    e = Ref{Float64}(0.0); v = Ref{Float64}(0.0); ov = Ref{Int32}(0)
    check(ccall((:mmc_ewald_short, libmmc), Int32,
                (Ptr{Cvoid}, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                s.ctx, i, sim_props.qq_rcut, e, v, ov))
    return e[], v[], ov[] != 0
end

function RecipLong(ewald, r::Vector{SVector{3,Float64}}, qq_q::Vector{Float64}, box::Float64)
    s = session(); bind_ewald!(s, ewald, box)
    GC.@preserve r begin                        # every atom matters here: re-send `r` (COM = NULL)
        check(ccall((:mmc_update_system, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                    s.ctx, C_NULL, pointer(r)))
    end
    s.last_mol = 0
    energy = Ref{Float64}(0.0)
    check(ccall((:mmc_recip_long, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}), s.ctx, energy))
    pull_s!(s, ewald; old = true)              # both arrays are written (ewalds.jl:600-601)
    return energy[], ewald
end

function RecipMove(box::Float64, ewalds, r_old::Vector, r_new::Vector, qq_q::Vector)
    s = session(); bind_ewald!(s, ewalds, box); push_s!(s, ewalds)
    ro = Vector{SVector{3,Float64}}(r_old); rn = Vector{SVector{3,Float64}}(r_new)
    q = Vector{Float64}(qq_q)
    de = Ref{Float64}(0.0)
    GC.@preserve ro rn q begin
        check(ccall((:mmc_recip_move, libmmc), Int32,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}),
                    s.ctx, pointer(ro), pointer(rn), pointer(q), length(q), de))
    end
    pull_s!(s, ewalds)                          # sumQExpNew was updated in place (:805-814)
    return de[], ewalds
end

function EwaldSelf(ewald, qq_q::Vector)
    s = session()
    e = Ref{Float64}(0.0)
    check(ccall((:mmc_ewald_self, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}), s.ctx, e))
    return e[]
end

function _fill!(tot, t::MMCTotals)
    tot.energy += t.energy; tot.virial += t.virial; tot.coulomb += t.coulomb
    return tot
end

"potential(moa, soa, tot, ewalds, vdwTable, sim_props, \"ewald\")   Ewald/energy.jl:946-1032"
function potential(moa, soa, tot, ewalds, vdwTable, sim_props, coulomb_style::String)
    s = session(); bind_ewald!(s, ewalds, sim_props.box); sync_all!(s, moa, soa)
    t = Ref{MMCTotals}()
    check(ccall((:mmc_potential_ewald, libmmc), Int32, (Ptr{Cvoid}, Float64, Float64, Ptr{MMCTotals}),
                s.ctx, sim_props.LJ_rcut, sim_props.qq_rcut, t))
    pull_s!(s, ewalds; old = true)
    return _fill!(tot, t[])
end

"potential(moa, soa, tot, ewald, vdwTable, sim_props)   (Wolf)   Ewald/energy.jl:864-943"
function potential(moa, soa, tot, ewald, vdwTable, sim_props)
    s = session(); bind_ewald!(s, ewald, sim_props.box); sync_all!(s, moa, soa)
    t = Ref{MMCTotals}()
    check(ccall((:mmc_potential_wolf, libmmc), Int32, (Ptr{Cvoid}, Float64, Float64, Ptr{MMCTotals}),
                s.ctx, sim_props.LJ_rcut, sim_props.qq_rcut, t))
    return _fill!(tot, t[])
end

export attach!, detach!, PrepareEwaldVariables, LJ_poly_ΔU, EwaldReal, EwaldShort, RecipLong,
       RecipMove, EwaldSelf, potential

end # module
