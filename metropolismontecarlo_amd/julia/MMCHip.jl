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
# ONE ccall PER REFERENCE CALL.  The per-molecule methods hand the library the caller's own arrays
# (mmc_call_lj_poly_du / mmc_call_ewald_short / mmc_call_ewald_real / mmc_call_recip_move,
# include/mmc_hip.h): the library compares molecule i and the molecule of the previous call with its
# mirror, evaluates LJ and real-space Coulomb together on the context's persistent kernel, answers
# the second of LJ_poly_ΔU(i) / EwaldShort(i) from that evaluation, computes RecipMove with the
# evaluation of the moved molecule, and finds out by content which of its buffers
# ewald.sumQExpOld / sumQExpNew are.  Measured through the Python mirror of this file (api.py, the
# same library calls): 33 us for the five calls of one Loop() iteration at 750 molecules, against
# 100 us for a C port of the reference on one core (bench.py, `call_surface`).
#
# BEYOND THE REFERENCE'S SURFACE (module MMCHipCore, nothing of Main is touched):
#   MMCHipCore.trial_move / accept_move! / reject_move!   the five calls of one Loop() iteration as
#       ONE evaluation (mmc_trial_move, mmc_accept_move, mmc_reject_move)
#   MMCHipCore.Batch                                      R independent replicas of the system on
#       one GPU (mmc_batch_*): north_star's "many independent NVT replicas fill the device" with
#       accept/reject on the Julia host (eval! / settle!) or in the library's native driver
#       (run! / run_chains!); see INTEGRATION.md.
# tests/test_julia_binding.py parses include/mmc_hip.h and every ccall below and compares symbol,
# arity and the C-to-Julia type of every argument, and the field layouts of the structs.
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

# mirrors of the structs of include/mmc_hip.h (field order, types and padding are checked by
# tests/test_julia_binding.py against the header)
struct MMCTotals
    energy::Float64; virial::Float64; coulomb::Float64
    lj::Float64; real::Float64; recip::Float64; self::Float64
    n_overlap::Int32; _pad::Int32
end

struct MMCMove
    mol::Int32; accept_prev::Int32
    com_new::NTuple{3,Float64}
    atoms_new::NTuple{9,Float64}
end

struct MMCMoveResult
    d_lj::Float64; d_real::Float64; d_recip::Float64; d_vir::Float64
    overlap::Int32; _pad::Int32
end

struct MMCRunParams
    temperature::Float64; dr_max::Float64; dphi_max::Float64
    seed::UInt64; n_steps::Int64
    n_groups::Int32; n_parts::Int32; time_kernels::Int32; n_threads::Int32; n_streams::Int32; _pad::Int32
    replica0::UInt64
end

struct MMCRunStats
    moves::Int64; launches::Int64
    trans_attempt::Int64; trans_accept::Int64; rot_attempt::Int64; rot_accept::Int64; overlaps::Int64
    wall_ms::Float64; kernel_ms::Float64; energy_sum::Float64
    timed_launches::Int64; torn_records::Int64; server_steps::Int64; device_decisions::Int64
end

struct MMCNptParams
    pressure::Float64; vmax::Float64; alpha::Float64
    n_sweeps::Int64; moves_per_sweep::Int64
end

struct MMCNptStats
    vol_attempt::Int64; vol_accept::Int64
    box::Float64; volume_sum::Float64; volume_ms::Float64
end

struct MMCChain
    dr_max::Float64; dphi_max::Float64
    energy::Float64; virial::Float64
    avg_energy::Float64; avg_virial::Float64
    steps_taken::Int64; overlaps::Int64
    trans_naccepp::Int64; trans_attempp::Int64; trans_naccept::Int64; trans_attempt::Int64
    rot_naccepp::Int64; rot_attempp::Int64; rot_naccept::Int64; rot_attempt::Int64
    trans_set_value::Float64; rot_set_value::Float64
end

"Address of the first Float64 of a Vector of bits types (SVector{3,Float64}, ComplexF64, Float64)."
fptr(x) = Ptr{Float64}(pointer(x))

mutable struct Session
    ctx::Ptr{Cvoid}
    box::Float64
    ewald_key::Tuple
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
                    ctx[], length(com), length(coords), fptr(com), pointer(fa), pointer(la),
                    fptr(coords), pointer(atype), pointer(charge), size(eps, 1), pointer(eps),
                    pointer(sig), box))
    end
    SESSION[] = Session(ctx[], box, ())
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

"Re-send every centre of mass and atom (after host edits that are not Loop()'s one-molecule pattern)."
function sync_all!(s::Session, com, coords)
    GC.@preserve com coords begin
        check(ccall((:mmc_update_system, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                    s.ctx, com === nothing ? Ptr{Float64}(C_NULL) : fptr(com), fptr(coords)))
    end
end

function bind_ewald!(s::Session, kappa, nk, k_sq_max, factor, box)
    key = (kappa, nk, k_sq_max, factor, box)
    if s.ewald_key != key
        n = Ref{Int64}(0)
        check(ccall((:mmc_prepare_ewald, libmmc), Int32,
                    (Ptr{Cvoid}, Float64, Int64, Int64, Float64, Float64, Ptr{Int64}),
                    s.ctx, kappa, nk, k_sq_max, box, factor, n))
        s.ewald_key = key
    end
end
bind_ewald!(s::Session, ewald, box) =
    bind_ewald!(s, ewald.kappa, ewald.nk, ewald.k_sq_max, ewald.factor, box)

"ewald.sumQExpOld / sumQExpNew <- the device's (RecipLong and potential write both, ewalds.jl:600-601)."
function pull_s!(s::Session, ewald; old::Bool = false)
    so, sn = ewald.sumQExpOld, ewald.sumQExpNew
    GC.@preserve so sn begin
        check(ccall((:mmc_get_sumqexp, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                    s.ctx, old ? fptr(so) : Ptr{Float64}(C_NULL), fptr(sn)))
    end
end

# ---- the per-molecule calls with the caller's own arrays: one ccall each ----
function call_lj_poly_du(s::Session, i::Int64, com, coords, r_cut::Float64)
    pot = Ref{Float64}(0.0); vir = Ref{Float64}(0.0)
    GC.@preserve com coords begin
        check(ccall((:mmc_call_lj_poly_du, libmmc), Int32,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}),
                    s.ctx, i, fptr(com), fptr(coords), r_cut, pot, vir))
    end
    return pot[], vir[]
end

function call_ewald_real(s::Session, i::Int64, com, coords, r_cut::Float64, ovr::Float64)
    pot = Ref{Float64}(0.0); ov = Ref{Int32}(0)
    GC.@preserve com coords begin
        check(ccall((:mmc_call_ewald_real, libmmc), Int32,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ptr{Float64}, Ptr{Int32}),
                    s.ctx, i, fptr(com), fptr(coords), r_cut, ovr, pot, ov))
    end
    return pot[], ov[] != 0
end

function call_ewald_short(s::Session, i::Int64, com, coords, qq_rcut::Float64)
    e = Ref{Float64}(0.0); v = Ref{Float64}(0.0); ov = Ref{Int32}(0)
    GC.@preserve com coords begin
        check(ccall((:mmc_call_ewald_short, libmmc), Int32,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}),
                    s.ctx, i, fptr(com), fptr(coords), qq_rcut, e, v, ov))
    end
    return e[], v[], ov[] != 0
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

# ---- one Loop() iteration as ONE evaluation (main.jl:491-593) ----
"""
    d, overlap = trial_move(i, com_new, atoms_new, lj_rcut, qq_rcut)

The five hot-path calls of one Loop() iteration for molecule `i` moved to `com_new` /
`atoms_new` (3 atoms): d = (ΔLJ, Δreal, ΔRecip, Δvirial) as `partial_new - partial_old` builds
them (main.jl:593,600-601).  The device keeps the OLD state: follow with `accept_move!()`
(main.jl:598-621) or `reject_move!()` (:622-629).  The host arrays are the caller's business.
"""
function trial_move(i::Int64, com_new::SVector{3,Float64}, atoms_new::Vector{SVector{3,Float64}},
                    lj_rcut::Float64, qq_rcut::Float64)
    s = session()
    d = zeros(Float64, 4); ov = Ref{Int32}(0); cn = [com_new]
    GC.@preserve cn atoms_new d begin
        check(ccall((:mmc_trial_move, libmmc), Int32,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ptr{Float64}, Ptr{Int32}),
                    s.ctx, i, fptr(cn), fptr(atoms_new), lj_rcut, qq_rcut, pointer(d), ov))
    end
    return (d[1], d[2], d[3], d[4]), ov[] != 0
end
accept_move!() = check(ccall((:mmc_accept_move, libmmc), Int32, (Ptr{Cvoid},), session().ctx))
reject_move!() = check(ccall((:mmc_reject_move, libmmc), Int32, (Ptr{Cvoid},), session().ctx))

"Counters of the context (mmc_ctx_stats): commands served, launches, retries, cache hits, ..."
function stats()
    out = zeros(Int64, 10)
    check(ccall((:mmc_ctx_stats, libmmc), Int32, (Ptr{Cvoid}, Ptr{Int64}), session().ctx, out))
    return out
end

# =================================================================================================
# The replica batch: R independent NVT chains of the attached kind of system on one GPU
# (include/mmc_hip.h, "replica batch").  Molecules must have 3 atoms (RecipMove, ewalds.jl:740).
# =================================================================================================
mutable struct Batch
    h::Ptr{Cvoid}
    n_replicas::Int64
    n_mol::Int64
end

"""
    b = Batch(n_replicas, moa, soa, vdwTable, ewald, box, lj_rcut, qq_rcut; device = 0)

Every replica starts from (moa, soa); `ewald` supplies kappa, nk, k_sq_max and factor
(main.jl:285-303).  Call `potential_ewald(b)` (or `recip_long!(b)`) once before the first move: it
fills the structure factors.
"""
function Batch(n_replicas::Integer, moa, soa, vdwTable, ewald, box::Float64, lj_rcut::Float64,
               qq_rcut::Float64; device::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    com, coords, atype, charge = moa.COM, soa.coords, soa.atype, soa.charge
    eps, sig = vdwTable.ϵᵢⱼ, vdwTable.σᵢⱼ
    GC.@preserve com coords atype charge eps sig begin
        check(ccall((:mmc_batch_create, libmmc), Int32,
                    (Int32, Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Int64},
                     Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Int64, Int64,
                     Float64, Float64, Float64, Ptr{Ptr{Cvoid}}),
                    device, C_NULL, n_replicas, length(com), fptr(com), fptr(coords), pointer(atype),
                    pointer(charge), size(eps, 1), pointer(eps), pointer(sig), box, ewald.kappa,
                    ewald.nk, ewald.k_sq_max, ewald.factor, lj_rcut, qq_rcut, h))
    end
    b = Batch(h[], n_replicas, length(com))
    finalizer(close!, b)
    return b
end

function close!(b::Batch)
    b.h == C_NULL && return
    ccall((:mmc_batch_destroy, libmmc), Int32, (Ptr{Cvoid},), b.h)
    b.h = C_NULL
    return
end

"Tuning switches of include/mmc_hip.h (\"kernel\", \"parts\", \"device_moves\", \"persistent\", ...)."
set_option!(b::Batch, key::String, value::Integer) =
    check(ccall((:mmc_batch_set_option, libmmc), Int32, (Ptr{Cvoid}, Cstring, Int64), b.h, key, value))

"potential(..., \"ewald\") of every replica (energy.jl:946-1032)."
function potential_ewald(b::Batch)
    tot = Vector{MMCTotals}(undef, b.n_replicas)
    check(ccall((:mmc_batch_potential_ewald, libmmc), Int32, (Ptr{Cvoid}, Ptr{MMCTotals}), b.h, tot))
    return tot
end

"RecipLong of every replica; energies WITHOUT factor (ewalds.jl:603)."
function recip_long!(b::Batch)
    e = Vector{Float64}(undef, b.n_replicas)
    check(ccall((:mmc_batch_recip_long, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}), b.h, e))
    return e
end

"""
    eval!(b, moves, results)

One trial move per replica in one launch: `moves[r]` (molecule, new COM and atoms;
`accept_prev` settles the replica's previous proposal first, main.jl:598-629) -> `results[r]`
(ΔLJ, Δreal, ΔRecip, Δvirial, overlap).  Accept/reject stays with the caller: this is Loop()'s
body for R chains at once.
"""
function eval!(b::Batch, moves::Vector{MMCMove}, results::Vector{MMCMoveResult})
    length(moves) == b.n_replicas == length(results) || error("one move and one result per replica")
    check(ccall((:mmc_batch_eval, libmmc), Int32, (Ptr{Cvoid}, Ptr{MMCMove}, Ptr{MMCMoveResult}),
                b.h, moves, results))
    return results
end

"Settle the last proposals (accept[r] != 0: commit, main.jl:598-621) without evaluating new ones."
function settle!(b::Batch, accept::Vector{Int32})
    length(accept) == b.n_replicas || error("one flag per replica")
    check(ccall((:mmc_batch_settle, libmmc), Int32, (Ptr{Cvoid}, Ptr{Int32}), b.h, accept))
end

"""
    stats = run!(b, params, energies)

The library's native driver: Loop()'s sequential accept/reject (main.jl:487-644) for every
replica, `params.n_steps` trial moves each; `energies[r]` is the running total.energy (:599).
"""
function run!(b::Batch, params::MMCRunParams, energies::Vector{Float64})
    length(energies) == b.n_replicas || error("one energy per replica")
    p = Ref(params); st = Ref{MMCRunStats}()
    check(ccall((:mmc_batch_run, libmmc), Int32,
                (Ptr{Cvoid}, Ptr{MMCRunParams}, Ptr{Float64}, Ptr{MMCRunStats}), b.h, p, energies, st))
    return st[]
end

"As `run!`, every chain with its own step sizes, block accumulators and Adjust! (adjust.jl:1-83)."
function run_chains!(b::Batch, params::MMCRunParams, chains::Vector{MMCChain}, adjust::Bool)
    length(chains) == b.n_replicas || error("one chain record per replica")
    p = Ref(params); st = Ref{MMCRunStats}()
    check(ccall((:mmc_batch_run_chains, libmmc), Int32,
                (Ptr{Cvoid}, Ptr{MMCRunParams}, Ptr{MMCChain}, Int32, Ptr{MMCRunStats}),
                b.h, p, chains, adjust ? 1 : 0, st))
    return st[]
end

"Coordinates (and sumQExpOld) of replica r (0-based, like the library)."
function get_replica(b::Batch, r::Integer)
    com = Vector{SVector{3,Float64}}(undef, b.n_mol)
    coords = Vector{SVector{3,Float64}}(undef, 3 * b.n_mol)
    s_old = Vector{ComplexF64}(undef, 337)
    GC.@preserve com coords s_old begin
        check(ccall((:mmc_batch_get_replica, libmmc), Int32,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    b.h, r, fptr(com), fptr(coords), fptr(s_old)))
    end
    return com, coords, s_old
end

function set_replica!(b::Batch, r::Integer, com::Vector{SVector{3,Float64}},
                      coords::Vector{SVector{3,Float64}})
    GC.@preserve com coords begin
        check(ccall((:mmc_batch_set_replica, libmmc), Int32,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}), b.h, r, fptr(com), fptr(coords)))
    end
end

"The device part of an NPT volume move for every replica (volumeChange.jl:59-80)."
volume_change!(b::Batch, new_box::Float64, new_kappa::Float64) =
    check(ccall((:mmc_batch_volume_change, libmmc), Int32, (Ptr{Cvoid}, Float64, Float64),
                b.h, new_box, new_kappa))

"""
    tot = volume_trial!(b, new_box, new_kappa); ...; volume_accept!(b) | volume_reject!(b)

An NPT volume move of a ONE-replica batch without a host round trip (Ewald/volumeChange.jl:59-147):
device-side snapshot, rescale, new tables, total energy at the new volume; a rejection restores
everything bit for bit.
"""
function volume_trial!(b::Batch, new_box::Float64, new_kappa::Float64)
    t = Ref{MMCTotals}()
    check(ccall((:mmc_batch_volume_trial, libmmc), Int32, (Ptr{Cvoid}, Float64, Float64, Ptr{MMCTotals}),
                b.h, new_box, new_kappa, t))
    return t[]
end
volume_accept!(b::Batch) = check(ccall((:mmc_batch_volume_accept, libmmc), Int32, (Ptr{Cvoid},), b.h))
volume_reject!(b::Batch) = check(ccall((:mmc_batch_volume_reject, libmmc), Int32, (Ptr{Cvoid},), b.h))

"""
    (run_stats, npt_stats) = run_npt!(b, params, npt, energy)

An NPT chain of a one-replica batch: `npt.n_sweeps` times { `npt.moves_per_sweep` trial moves
(Loop(), main.jl:487-644), one volume move (volumeChange.jl:59-147) }; `energy[1]` is the running
total.energy of the replica.
"""
function run_npt!(b::Batch, params::MMCRunParams, npt::MMCNptParams, energy::Vector{Float64})
    length(energy) == 1 || error("an NPT chain is a batch of one replica")
    p = Ref(params); q = Ref(npt); st = Ref{MMCRunStats}(); ns = Ref{MMCNptStats}()
    check(ccall((:mmc_batch_run_npt, libmmc), Int32,
                (Ptr{Cvoid}, Ptr{MMCRunParams}, Ptr{MMCNptParams}, Ptr{Float64}, Ptr{MMCRunStats},
                 Ptr{MMCNptStats}), b.h, p, q, energy, st, ns))
    return st[], ns[]
end

# ---- the one collective of a sharded run: RCCL behind the C ABI (include/mmc_hip.h, mmc_dist_*) ----
"""
    id = dist_unique_id()                       # rank 0; send the 128 bytes to the other ranks
    d  = dist_init(rank, world, id; device = rank)
    dist_reduce!(d, sums, maxima)               # in place over all ranks: sum / max, Float64
    dist_destroy(d)

One process per GPU; replicas shard over ranks with no data-path collective, and this is the
reduction of a block's observables (energy sums, acceptance counters; the longest elapsed time).
"""
function dist_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:mmc_dist_unique_id, libmmc), Int32, (Ptr{UInt8},), id))
    return id
end
function dist_init(rank::Integer, world::Integer, id::Vector{UInt8}; device::Integer = 0)
    length(id) == 128 || error("the unique id has 128 bytes")
    d = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:mmc_dist_init, libmmc), Int32, (Int32, Int32, Ptr{UInt8}, Int32, Ptr{Ptr{Cvoid}}),
                rank, world, id, device, d))
    return d[]
end
dist_reduce!(d::Ptr{Cvoid}, sums::Vector{Float64}, maxima::Vector{Float64}) =
    check(ccall((:mmc_dist_reduce, libmmc), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}, Int64),
                d, sums, length(sums), maxima, length(maxima)))
dist_destroy(d::Ptr{Cvoid}) = check(ccall((:mmc_dist_destroy, libmmc), Int32, (Ptr{Cvoid},), d))

"Status line of one block as Loop() prints it (main.jl:667-679) from one chain record."
function block_line(chain::MMCChain, block::Integer, n_mol::Integer, box::Float64;
                    ideal_term::Float64 = 4.60453)
    buf = Vector{UInt8}(undef, 512); c = Ref(chain)
    check(ccall((:mmc_chain_block_line, libmmc), Int32,
                (Ptr{MMCChain}, Int64, Int64, Float64, Float64, Ptr{UInt8}, Int64),
                c, block, n_mol, box, ideal_term, buf, length(buf)))
    return unsafe_string(pointer(buf))
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
                      ctx[], Ptr{Int32}(pointer(kxyz)), pointer(cfac)))
    end
    ccall((:mmc_ctx_destroy, C.libmmc), Int32, (Ptr{Cvoid},), ctx[])
    return EWALD(ewald.kappa, ewald.nk, ewald.k_sq_max, oftype(ewald.nk, n[]), kxyz, cfac,
                 zeros(ComplexF64, n[]), zeros(ComplexF64, n[]), ewald.factor)   # ewalds.jl:91-101
end

# ---- Ewald/energy.jl:209-210 ----------------------------------------------------------------------
function LJ_poly_ΔU(i, moa::StructArray, soa::StructArray,
                            vdwTable, r_cut, box)
    return MMCHipCore.call_lj_poly_du(MMCHipCore.session(), Int64(i), moa.COM, soa.coords, Float64(r_cut))
end

# ---- Ewald/energy.jl:126 (legacy) -----------------------------------------------------------------
function LJ_poly_ΔU(i::Int, system::Requirements)
    return MMCHipCore.call_lj_poly_du(MMCHipCore.session(), Int64(i), system.rm, system.ra, system.r_cut)
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
    return MMCHipCore.call_ewald_real(s, chosenOne, moa.COM, soa.coords, r_cut, 0.5)  # ovr = 0.5 (ewalds.jl:327)
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
    return MMCHipCore.call_ewald_real(s, chosenOne, system.rm, qq_r, system.r_cut, 1.0)  # ovr = 1.0 (ewalds.jl:240)
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
    s = MMCHipCore.session(); MMCHipCore.bind_ewald!(s, ewald, box)
    return MMCHipCore.call_ewald_short(s, i, moa.COM, soa.coords, sim_props.qq_rcut)
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
    s = C.session(); C.bind_ewald!(s, ewalds, box)
    ro = Vector{SVector{3,Float64}}(r_old); rn = Vector{SVector{3,Float64}}(r_new)
    q = Vector{Float64}(qq_q)
    so, sn = ewalds.sumQExpOld, ewalds.sumQExpNew   # as they are NOW: Loop rebinds them (main.jl:621,628)
    de = Ref{Float64}(0.0)
    GC.@preserve ro rn q so sn begin
        C.check(ccall((:mmc_call_recip_move, C.libmmc), Int32,
                      (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Float64},
                       Ptr{Float64}, Ptr{Float64}),
                      s.ctx, C.fptr(ro), C.fptr(rn), pointer(q), length(q), C.fptr(so), C.fptr(sn), de))
    end
    return de[], ewalds                         # sumQExpNew was updated in place (:805-814)
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
    s = C.session(); C.sync_all!(s, system.rm, qq_r)
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
