"""The Julia binding (metropolismontecarlo_amd/julia/MMCHip.jl) cannot be executed here: the
image has no `julia`.  What can be checked by text is the property that makes it take effect --
every hot-path method of the reference is redefined at TOP LEVEL (outside any module, so in the
module that includes the file: Main) with exactly the reference's type signature, so that Julia
overwrites the CPU method instead of adding a less specific one beside it.

The expected signatures below are the reference's, normalised (whitespace and comments removed);
each cites where it stands in /root/reference/Ewald."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
JL = os.path.join(os.path.dirname(HERE), "metropolismontecarlo_amd", "julia", "MMCHip.jl")

SIGNATURES = {
    # current moa / soa API
    "energy.jl:209-210": "LJ_poly_ΔU(i,moa::StructArray,soa::StructArray,vdwTable,r_cut,box)",
    "ewalds.jl:293-299": "EwaldReal(chosenOne::Int64,moa::StructArray,soa::StructArray,ewald::EWALD,"
                         "r_cut::Float64,box::Float64)",
    "ewalds.jl:892-899": "EwaldShort(i::Int64,moa::StructArray,soa::StructArray,sim_props::Properties2,"
                         "ewald::EWALD,box::Float64,)",
    "ewalds.jl:45": "PrepareEwaldVariables(ewald::EWALD,boxSize::Realwhere{T})",
    "ewalds.jl:538-543": "RecipLong(ewald::EWALD,r::Vector{SVector{3,Float64}},qq_q::Vector{Float64},"
                         "box::Float64)",
    "ewalds.jl:718-724": "RecipMove(box::Float64,ewalds::EWALD,r_old::Vector,r_new::Vector,qq_q::Vector,)",
    "ewalds.jl:829": "EwaldSelf(ewald::EWALD,qq_q::Vector)",
    "energy.jl:946-954": "potential(moa::StructArray,soa::StructArray,tot::Properties,ewalds::EWALD,"
                         "vdwTable::Tables,sim_props::Properties2,coulomb_style::String)",
    "energy.jl:864-871": "potential(moa::StructArray,soa::StructArray,tot::Properties,ewald::EWALD,"
                         "vdwTable::Tables,sim_props::Properties2)",
    # legacy Requirements API
    "energy.jl:126": "LJ_poly_ΔU(i::Int,system::Requirements)",
    "ewalds.jl:205-213": "EwaldReal(qq_r::Vector{SVector{3,Float64}},qq_q::Vector{Float64},kappa::Real,"
                         "box::Float64,thisMol_thisAtom::Vector{SVector{2,Int64}},chosenOne::Int64,"
                         "system::Requirements,)",
    "ewalds.jl:848-856": "EwaldShort(i::Int64,system::Requirements,ewald::EWALD,box::Float64,"
                         "qq_r::Vector{SVector{3,Float64}},qq_q::Vector{Float64},tinfoil=false,)",
    "ewalds.jl:465-470": "RecipLong(system::Requirements,ewald::EWALD,r::Vector{SVector{3,Float64}},"
                         "qq_q::Vector{Float64},)",
    "energy.jl:618-624": "CoulombReal(qq_r::Vector{SVector{3,Float64}},qq_q::Vector{Float64},box::Float64,"
                         "chosenOne::Int64,system::Requirements)",
}


def top_level_signatures(text):
    """`function name(args)` headers that are not nested inside a `module ... end` block,
    normalised like SIGNATURES."""
    # drop the core module (its body is indented code between `module MMCHipCore` and its `end`)
    text = re.sub(r"(?ms)^module MMCHipCore\b.*?^end # module MMCHipCore\s*$", "", text)
    assert "module " not in re.sub(r"(?m)^\s*#.*$", "", text), "methods must not live in a module"
    out = []
    for m in re.finditer(r"(?ms)^function\s+(\S+?)\((.*?)\n?\s*\)\s*$", text):
        args = re.sub(r"#[^\n]*", "", m.group(2))          # the reference's trailing comments
        out.append(re.sub(r"\s+", "", f"{m.group(1)}({args})"))
    return out


def test_every_reference_method_is_overwritten_with_its_exact_signature():
    got = top_level_signatures(open(JL, encoding="utf-8").read())
    assert len(got) == len(set(got)), "a signature is defined twice"
    want = {re.sub(r"\s+", "", v): k for k, v in SIGNATURES.items()}
    missing = [f"{want[w]}: {w}" for w in want if w not in got]
    assert not missing, "not redefined with the reference's signature:\n" + "\n".join(missing)
    extra = [g for g in got if g not in want]
    assert not extra, f"top-level methods the reference does not have: {extra}"


def test_signatures_are_the_references_own(tmp_path):
    """Where the reference tree is present (the build container, not the GPU box) the expected
    signatures above are compared with the reference's source text itself."""
    ref = "/root/reference/Ewald"
    if not os.path.isdir(ref):
        import pytest
        pytest.skip("reference tree not present on this machine")
    for where, sig in SIGNATURES.items():
        fname, lines = where.split(":")
        lo = int(lines.split("-")[0])
        src = open(os.path.join(ref, fname), encoding="utf-8").read().split("\n")
        # the header starts at line `lo` and runs to the line that closes the argument list
        head = ""
        for ln in src[lo - 1:lo + 12]:
            head += re.sub(r"#[^\n]*", "", ln) + "\n"
            if head.count("(") and head.count("(") == head.count(")"):
                break
        norm = re.sub(r"\s+", "", head)
        assert norm.startswith("function" + re.sub(r"\s+", "", sig)), (where, norm, sig)


def test_binding_says_how_it_takes_effect_and_that_it_was_not_run():
    text = open(JL, encoding="utf-8").read()
    assert "include" in text and "AFTER" in text and "overwrites" in text
    assert "NOT RUN" in text and "no `julia`" in text
    integ = open(os.path.join(os.path.dirname(HERE), "INTEGRATION.md"), encoding="utf-8").read()
    assert "using .MMCHip" not in integ                     # the recipe that silently did nothing
    assert "MMCHipCore.attach!" in integ and "has not been run" in integ
