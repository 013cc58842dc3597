# MMCHip.jl -- the reference's hot-path methods served by libmmc_hip.so (MI355X / gfx950).
#
# HOW IT TAKES EFFECT.  The reference is a script: `Ewald/main.jl` `include`s energy.jl, ewalds.jl,
# auxillary.jl, structs.jl into `Main`, so `LJ_poly_ΔU`, `EwaldReal`, ... are generic functions OWNED
# BY `Main`.  A module that exports functions of the same names cannot replace them (`using` of a
# name Main already owns is a conflict and Main's CPU methods keep being called), and a method with
# a looser signature loses dispatch to the reference's typed one.  Therefore this file is
# `include`d INTO `Main`, AFTER the reference's own includes, and
#   * keeps its session state and ccall helpers inside `module MMCHipCore` (no name clashes), and
#   * defines, at top level, one method per reference method WITH THE REFERENCE'S EXACT TYPE
#     SIGNATURE (each is quoted with its file:line): Julia then overwrites the CPU method in place
#     -- same function object, same signature, new body -- and every existing caller (`Loop()`,
#     `potential`, the tests) runs the GPU path without being edited.
#
#     include("energy.jl"); include("ewalds.jl"); ...            # the reference, unchanged
#     include("/path/to/metropolismontecarlo_amd/julia/MMCHip.jl")
#     MMCHipCore.attach!(moa, soa, vdwTable, box)     # once, after MakeAtomArrays / MakeTables
#       (legacy API: MMCHipCore.attach!(system::Requirements, qq_q))
#
# Overwritten (current moa/soa API)                                     reference
#   LJ_poly_ΔU(i, moa::StructArray, soa::StructArray, vdwTable, r_cut, box)   Ewald/energy.jl:209-210
#   EwaldReal(chosenOne::Int64, moa::StructArray, soa::StructArray, ewald::EWALD,
#             r_cut::Float64, box::Float64)                                   Ewald/ewalds.jl:293-299
#   EwaldShort(i::Int64, moa::StructArray, soa::StructArray, sim_props::Properties2,
#              ewald::EWALD, box::Float64)                                    Ewald/ewalds.jl:892-899
#   PrepareEwaldVariables(ewald::EWALD, boxSize::Real where {T})              Ewald/ewalds.jl:45
#   RecipLong(ewald::EWALD, r::Vector{SVector{3,Float64}}, qq_q::Vector{Float64},
#             box::Float64)                                                   Ewald/ewalds.jl:538-543
#   RecipMove(box::Float64, ewalds::EWALD, r_old::Vector, r_new::Vector, qq_q::Vector)
#                                                                             Ewald/ewalds.jl:718-724
#   EwaldSelf(ewald::EWALD, qq_q::Vector)                                     Ewald/ewalds.jl:829
#   potential(moa::StructArray, soa::StructArray, tot::Properties, ewalds::EWALD, vdwTable::Tables,
#             sim_props::Properties2, coulomb_style::String)                  Ewald/energy.jl:946-954
#   potential(moa::StructArray, soa::StructArray, tot::Properties, ewald::EWALD, vdwTable::Tables,
#             sim_props::Properties2)                       (Wolf)            Ewald/energy.jl:864-871
# Overwritten (legacy `Requirements` API)
#   LJ_poly_ΔU(i::Int, system::Requirements)                                  Ewald/energy.jl:126
#   EwaldReal(qq_r::Vector{SVector{3,Float64}}, qq_q::Vector{Float64}, kappa::Real, box::Float64,
#             thisMol_thisAtom::Vector{SVector{2,Int64}}, chosenOne::Int64,
#             system::Requirements)                                           Ewald/ewalds.jl:205-213
#   EwaldShort(i::Int64, system::Requirements, ewald::EWALD, box::Float64,
#              qq_r::Vector{SVector{3,Float64}}, qq_q::Vector{Float64}, tinfoil = false)
#                                                                             Ewald/ewalds.jl:848-856
#   RecipLong(system::Requirements, ewald::EWALD, r::Vector{SVector{3,Float64}},
#             qq_q::Vector{Float64})                                          Ewald/ewalds.jl:465-470
#   CoulombReal(qq_r::Vector{SVector{3,Float64}}, qq_q::Vector{Float64}, box::Float64,
#               chosenOne::Int64, system::Requirements)                       Ewald/energy.jl:618-624
# Return values are the reference's: (pot, vir), (pot, overlap::Bool), (e, v, overlap),
# (energy, ewald), ΔE, energy, tot.  Host arrays are borrowed for the duration of each ccall
# (GC.@preserve); nothing is cached by pointer (Loop() rebinds ewald.sumQExpOld/New on every move,
# main.jl:621,628).
#
# NOT RUN.  The build image has no `julia` (and no network to fetch one), so this file has never
# been executed; it is written against the reference's source and the C header it binds
# (include/mmc_hip.h), and tests/test_julia_binding.py checks by text that every method above is
# present here with the reference's signature line.  The same C ABI is exercised on the GPU by the
# Python mirror (metropolismontecarlo_amd/api.py).

for needed in (:EWALD, :Requirements, :Properties, :Properties2, :Tables, :StructArray, :SVector)
    isdefined(@__MODULE__, needed) ||
        error("MMCHip.jl must be included AFTER the reference's structs.jl, auxillary.jl, " *
              "energy.jl and ewalds.jl (and their `using StaticArrays, StructArrays`): " *
              "$needed is not defined in $(@__MODULE__)")
end

module MMCHipCore

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

function upload!(com, fa, la, coords, atype, charge, eps, sig, box::Float64, device::Integer)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mmc_ctx_create, libmmc), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                device, C_NULL, ctx))
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

"Upload moa/soa/vdwTable once (mmc_upload_system): the current API of Loop()."
attach!(moa, soa, vdwTable, box::Float64; device::Integer = 0) =
    upload!(moa.COM, moa.firstAtom, moa.lastAtom, soa.coords, soa.atype, soa.charge,
            vdwTable.ϵᵢⱼ, vdwTable.σᵢⱼ, box, device)

"Upload a legacy `Requirements` system (auxillary.jl:59-75) and its charges."
function attach!(system, qq_q::Vector{Float64}; device::Integer = 0)
    fa = Int64[t[1] for t in system.thisMol_theseAtoms]
    la = Int64[t[2] for t in system.thisMol_theseAtoms]
    atype = Vector{Int64}(system.atomTypes)
    return upload!(system.rm, fa, la, system.ra, atype, qq_q, system.table.ϵᵢⱼ, system.table.σᵢⱼ,
                   system.box, device)
end

function detach!()
    s = SESSION[]
    s === nothing && return
    ccall((:mmc_ctx_destroy, libmmc), Int32, (Ptr{Cvoid},), s.ctx)
    SESSION[] = nothing
end

session() = (s = SESSION[]; s === nothing ? error("MMCHipCore.attach!(...) first") : s)

# Loop() changes COM[i] / coords[first:last] of ONE molecule between calls and may have restored
# the previous one (main.jl:527,552,623-624): re-send both.
function sync_molecule!(s::Session, com, coords, first_atom, i::Int64)
    for m in unique((i, s.last_mol))
        m == 0 && continue
        f = first_atom(m)
        GC.@preserve com coords begin
            check(ccall((:mmc_set_molecule, libmmc), Int32,
                        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}),
                        s.ctx, m, pointer(com, m), pointer(coords, f)))
        end
    end
    s.last_mol = i
end
sync_molecule!(s::Session, moa, soa, i::Int64) =
    sync_molecule!(s, moa.COM, soa.coords, m -> moa.firstAtom[m], i)
sync_molecule_legacy!(s::Session, system, qq_r, i::Int64) =
    sync_molecule!(s, system.rm, qq_r, m -> system.thisMol_theseAtoms[m][1], i)

function sync_all!(s::Session, com, coords)
    GC.@preserve com coords begin
        check(ccall((:mmc_update_system, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                    s.ctx, com === nothing ? C_NULL : pointer(com), pointer(coords)))
    end
    s.last_mol = 0
end

function bind_ewald!(s::Session, kappa, nk, k_sq_max, factor, box)
    key = (kappa, nk, k_sq_max, factor, box)
    if s.ewald_key != key
        n = Ref{Int64}(0)
        check(ccall((:mmc_prepare_ewald, libmmc), Int32,
                    (Ptr{Cvoid}, Float64, Int64, Int64, Float64, Float64, Ptr{Int64}),
                    s.ctx, kappa, nk, k_sq_max, box, factor, n))
        s.ewald_key = key
        s.s_old = ComplexF64[]; s.s_new = ComplexF64[]   # device arrays were zeroed
    end
end
bind_ewald!(s::Session, ewald, box) =
    bind_ewald!(s, ewald.kappa, ewald.nk, ewald.k_sq_max, ewald.factor, box)

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

function lj_poly_du(s::Session, i::Int64, r_cut::Float64)
    pot = Ref{Float64}(0.0); vir = Ref{Float64}(0.0)
    check(ccall((:mmc_lj_poly_du, libmmc), Int32,
                (Ptr{Cvoid}, Int64, Float64, Ptr{Float64}, Ptr{Float64}), s.ctx, i, r_cut, pot, vir))
    return pot[], vir[]
end

function ewald_real(s::Session, i::Int64, r_cut::Float64, ovr::Float64)
    pot = Ref{Float64}(0.0); ov = Ref{Int32}(0)
    check(ccall((:mmc_ewald_real, libmmc), Int32,
                (Ptr{Cvoid}, Int64, Float64, Float64, Ptr{Float64}, Ptr{Int32}),
                s.ctx, i, r_cut, ovr, pot, ov))
    return pot[], ov[] != 0
end

function recip_long(s::Session, ewald)
    energy = Ref{Float64}(0.0)
    check(ccall((:mmc_recip_long, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}), s.ctx, energy))
    pull_s!(s, ewald; old = true)              # both arrays are written (ewalds.jl:600-601)
    return energy[]
end

function totals(s::Session, sym::Symbol, lj_rcut::Float64, qq_rcut::Float64)
    t = Ref{MMCTotals}()
    if sym === :ewald
        check(ccall((:mmc_potential_ewald, libmmc), Int32,
                    (Ptr{Cvoid}, Float64, Float64, Ptr{MMCTotals}), s.ctx, lj_rcut, qq_rcut, t))
    else
        check(ccall((:mmc_potential_wolf, libmmc), Int32,
                    (Ptr{Cvoid}, Float64, Float64, Ptr{MMCTotals}), s.ctx, lj_rcut, qq_rcut, t))
    end
    return t[]
end

end # module MMCHipCore

# =================================================================================================
# The reference's methods, redefined in the including module (Main) with their exact signatures.
# =================================================================================================

# ---- Ewald/ewalds.jl:45 ---------------------------------------------------------------------------
function PrepareEwaldVariables(ewald::EWALD, boxSize::Real where {T})
    C = MMCHipCore
    box = Float64(min(boxSize...))                       # ewalds.jl:50
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    C.check(ccall((:mmc_ctx_create, C.libmmc), Int32, (Int32, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}), 0, C_NULL, ctx))
    n = Ref{Int64}(0)
    st = ccall((:mmc_prepare_ewald, C.libmmc), Int32,
               (Ptr{Cvoid}, Float64, Int64, Int64, Float64, Float64, Ptr{Int64}),
               ctx[], ewald.kappa, ewald.nk, ewald.k_sq_max, box, ewald.factor, n)
    st != 0 && (ccall((:mmc_ctx_destroy, C.libmmc), Int32, (Ptr{Cvoid},), ctx[]); C.check(st))
    kxyz = Vector{SVector{3,Int32}}(undef, n[])
    cfac = Vector{Float64}(undef, n[])
    GC.@preserve kxyz cfac begin
        C.check(ccall((:mmc_get_kvectors, C.libmmc), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}),
                      ctx[], pointer(kxyz), pointer(cfac)))
    end
    ccall((:mmc_ctx_destroy, C.libmmc), Int32, (Ptr{Cvoid},), ctx[])
    return EWALD(ewald.kappa, ewald.nk, ewald.k_sq_max, oftype(ewald.nk, n[]), kxyz, cfac,
                 zeros(ComplexF64, n[]), zeros(ComplexF64, n[]), ewald.factor)   # ewalds.jl:91-101
end

# ---- Ewald/energy.jl:209-210 ----------------------------------------------------------------------
function LJ_poly_ΔU(i, moa::StructArray, soa::StructArray,
                            vdwTable, r_cut, box)
    s = MMCHipCore.session(); MMCHipCore.sync_molecule!(s, moa, soa, Int64(i))
    return MMCHipCore.lj_poly_du(s, Int64(i), Float64(r_cut))
end

# ---- Ewald/energy.jl:126 (legacy) -----------------------------------------------------------------
function LJ_poly_ΔU(i::Int, system::Requirements)
    s = MMCHipCore.session(); MMCHipCore.sync_molecule_legacy!(s, system, system.ra, Int64(i))
    return MMCHipCore.lj_poly_du(s, Int64(i), system.r_cut)
end

# ---- Ewald/ewalds.jl:293-299 ----------------------------------------------------------------------
function EwaldReal(chosenOne::Int64,
                    moa::StructArray,
                    soa::StructArray,
                    ewald::EWALD,
                    r_cut::Float64,
                    box::Float64
    )
    s = MMCHipCore.session(); MMCHipCore.bind_ewald!(s, ewald, box)
    MMCHipCore.sync_molecule!(s, moa, soa, chosenOne)
    return MMCHipCore.ewald_real(s, chosenOne, r_cut, 0.5)         # ovr = 0.5 (ewalds.jl:327)
end

# ---- Ewald/ewalds.jl:205-213 (legacy: ovr = 1.0, cutoff from system.r_cut) --------------------------
function EwaldReal(
    qq_r::Vector{SVector{3,Float64}},
    qq_q::Vector{Float64},
    kappa::Real,
    box::Float64,
    thisMol_thisAtom::Vector{SVector{2,Int64}},
    chosenOne::Int64,
    system::Requirements,
)
    s = MMCHipCore.session()
    MMCHipCore.bind_ewald!(s, Float64(kappa), 5, 27, factor, box)  # `factor`: constants.jl:28
    MMCHipCore.sync_molecule_legacy!(s, system, qq_r, chosenOne)
    return MMCHipCore.ewald_real(s, chosenOne, system.r_cut, 1.0)  # ovr = 1.0 (ewalds.jl:240)
end

# ---- Ewald/ewalds.jl:892-899 ----------------------------------------------------------------------
function EwaldShort(
    i::Int64,
    moa::StructArray,
    soa::StructArray,
    sim_props::Properties2,
    ewald::EWALD,
    box::Float64,
)
    C = MMCHipCore
    s = C.session(); C.bind_ewald!(s, ewald, box); C.sync_molecule!(s, moa, soa, i)
    e = Ref{Float64}(0.0); v = Ref{Float64}(0.0); ov = Ref{Int32}(0)
    C.check(ccall((:mmc_ewald_short, C.libmmc), Int32,
                  (Ptr{Cvoid}, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                  s.ctx, i, sim_props.qq_rcut, e, v, ov))
    return e[], v[], ov[] != 0
end

# ---- Ewald/ewalds.jl:848-856 (legacy) -------------------------------------------------------------
function EwaldShort(
    i::Int64,
    system::Requirements,
    ewald::EWALD,
    box::Float64,
    qq_r::Vector{SVector{3,Float64}},
    qq_q::Vector{Float64},
    tinfoil = false,
)
    realEwald, overlap = EwaldReal(qq_r, qq_q, ewald.kappa, box, system.thisMol_theseAtoms, i, system)
    realEwald *= ewald.factor                                      # ewalds.jl:870
    return realEwald, realEwald / 3, overlap                       # :871-872, :887
end

# ---- Ewald/ewalds.jl:538-543 ----------------------------------------------------------------------
function RecipLong(
    ewald::EWALD,
    r::Vector{SVector{3,Float64}},
    qq_q::Vector{Float64},
    box::Float64
)
    s = MMCHipCore.session(); MMCHipCore.bind_ewald!(s, ewald, box)
    MMCHipCore.sync_all!(s, nothing, r)        # every atom matters here: re-send `r` (COM = NULL)
    return MMCHipCore.recip_long(s, ewald), ewald
end

# ---- Ewald/ewalds.jl:465-470 (legacy: the box is system.box, :478) ---------------------------------
function RecipLong(
    system::Requirements,
    ewald::EWALD,
    r::Vector{SVector{3,Float64}},
    qq_q::Vector{Float64},
)
    s = MMCHipCore.session(); MMCHipCore.bind_ewald!(s, ewald, system.box)
    MMCHipCore.sync_all!(s, nothing, r)
    return MMCHipCore.recip_long(s, ewald), ewald
end

# ---- Ewald/ewalds.jl:718-724 ----------------------------------------------------------------------
function RecipMove(
    box::Float64,
    ewalds::EWALD,
    r_old::Vector,
    r_new::Vector,
    qq_q::Vector,
)
    C = MMCHipCore
    s = C.session(); C.bind_ewald!(s, ewalds, box); C.push_s!(s, ewalds)
    ro = Vector{SVector{3,Float64}}(r_old); rn = Vector{SVector{3,Float64}}(r_new)
    q = Vector{Float64}(qq_q)
    de = Ref{Float64}(0.0)
    GC.@preserve ro rn q begin
        C.check(ccall((:mmc_recip_move, C.libmmc), Int32,
                      (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64}),
                      s.ctx, pointer(ro), pointer(rn), pointer(q), length(q), de))
    end
    C.pull_s!(s, ewalds)                        # sumQExpNew was updated in place (:805-814)
    return de[], ewalds
end

# ---- Ewald/ewalds.jl:829 --------------------------------------------------------------------------
function EwaldSelf(ewald::EWALD, qq_q::Vector)
    s = MMCHipCore.session()
    e = Ref{Float64}(0.0)
    MMCHipCore.check(ccall((:mmc_ewald_self, MMCHipCore.libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}),
                           s.ctx, e))
    return e[]
end

# ---- Ewald/energy.jl:618-624 (bare Coulomb, legacy API only) ---------------------------------------
function CoulombReal(
    qq_r::Vector{SVector{3,Float64}},
    qq_q::Vector{Float64},
    box::Float64,
    chosenOne::Int64,
    system::Requirements
)
    C = MMCHipCore
    s = C.session(); C.sync_molecule_legacy!(s, system, qq_r, chosenOne)
    pot = Ref{Float64}(0.0); ov = Ref{Int32}(0)
    # `@assert r_cut == 10.0` (energy.jl:648) comes back as status 2 -> AssertionError
    C.check(ccall((:mmc_coulomb_real, C.libmmc), Int32,
                  (Ptr{Cvoid}, Int64, Float64, Ptr{Float64}, Ptr{Int32}),
                  s.ctx, chosenOne, system.r_cut, pot, ov))
    return pot[], ov[] != 0
end

# ---- Ewald/energy.jl:946-954 ----------------------------------------------------------------------
function potential(
    moa::StructArray,
    soa::StructArray,
    tot::Properties,
    ewalds::EWALD,
    vdwTable::Tables,
    sim_props::Properties2,
    coulomb_style::String #triggers wolf summations using double strings
)
    C = MMCHipCore
    s = C.session(); C.bind_ewald!(s, ewalds, sim_props.box); C.sync_all!(s, moa.COM, soa.coords)
    t = C.totals(s, :ewald, sim_props.LJ_rcut, sim_props.qq_rcut)
    C.pull_s!(s, ewalds; old = true)           # RecipLong inside wrote both arrays (energy.jl:1008)
    # the reference adds onto `tot` and halves what it holds after the LJ loop (energy.jl:964-969)
    tot.energy = tot.energy / 2 + t.energy; tot.virial = tot.virial / 2 + t.virial
    tot.coulomb += t.coulomb
    return tot
end

# ---- Ewald/energy.jl:864-871 (Wolf) ---------------------------------------------------------------
function potential(
    moa::StructArray,
    soa::StructArray,
    tot::Properties,
    ewald::EWALD,
    vdwTable::Tables,
    sim_props::Properties2#triggers wolf summations using double strings
)
    C = MMCHipCore
    s = C.session(); C.bind_ewald!(s, ewald, sim_props.box); C.sync_all!(s, moa.COM, soa.coords)
    t = C.totals(s, :wolf, sim_props.LJ_rcut, sim_props.qq_rcut)
    tot.energy = tot.energy / 2 + t.energy; tot.virial = tot.virial / 2 + t.virial   # energy.jl:882-887
    tot.coulomb += t.coulomb
    return tot
end
