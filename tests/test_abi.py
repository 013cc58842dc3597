"""CPU tests of the drop-in boundary: the C-ABI library loads and exports exactly what
include/mmc_hip.h declares; struct layouts match; without a GPU every compute entry point fails
loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from metropolismontecarlo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmc_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 35
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in mmc_hip.h but not exported by libmmc_hip.so"


def test_binding_covers_header_and_nothing_else():
    assert _lib.exported_symbols() == header_functions()
    _lib.lib()  # sets argtypes for all of them; AttributeError if one is missing


def test_no_cxx_or_torch_types_in_the_abi():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = [l.split()[-1] for l in out.splitlines() if " T " in l]
    mmc = [s for s in exported if s.startswith("mmc_")]
    assert sorted(mmc) == header_functions()
    src = open(HEADER).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "torch" not in code.lower() and "std::" not in code and "hipStream_t" not in code
    assert "#include <hip" not in code and "at::" not in code
    needed = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH], text=True)
    assert "libtorch" not in needed and "libc10" not in needed
    assert "liboracle" not in needed and "mmc_oracle" not in needed


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of every struct of the header, as gcc lays them out, against the ctypes
    and numpy mirrors the Python side passes by pointer."""
    prog = tmp_path / "layout.c"
    prog.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "mmc_hip.h"\n'
        "int main(void){\n"
        'printf("%zu %zu %zu\\n", sizeof(mmc_move), offsetof(mmc_move, com_new), offsetof(mmc_move, atoms_new));\n'
        'printf("%zu %zu\\n", sizeof(mmc_move_result), sizeof(mmc_totals));\n'
        'printf("%zu %zu %zu\\n", sizeof(mmc_run_params), offsetof(mmc_run_params, n_streams), offsetof(mmc_run_params, replica0));\n'
        'printf("%zu %zu %zu\\n", sizeof(mmc_run_stats), offsetof(mmc_run_stats, timed_launches), offsetof(mmc_run_stats, device_decisions));\n'
        'printf("%zu %zu\\n", sizeof(mmc_chain), offsetof(mmc_chain, trans_set_value));\n'
        "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.dirname(HEADER), str(prog), "-o", str(exe)])
    rows = [list(map(int, l.split())) for l in subprocess.check_output([str(exe)], text=True).splitlines()]
    assert rows[0] == [C.sizeof(_lib.Move), _lib.Move.com_new.offset, _lib.Move.atoms_new.offset]
    assert rows[0] == [8 + 24 + 72, 8, 32]
    assert rows[1] == [C.sizeof(_lib.MoveResult), C.sizeof(_lib.Totals)] == [40, 64]
    assert rows[1][1] == _lib.TOTALS_DTYPE.itemsize
    assert rows[2] == [C.sizeof(_lib.RunParams), _lib.RunParams.n_streams.offset,
                       _lib.RunParams.replica0.offset]
    assert rows[3] == [C.sizeof(_lib.RunStats), _lib.RunStats.timed_launches.offset,
                       _lib.RunStats.device_decisions.offset]
    assert rows[4] == [_lib.CHAIN_DTYPE.itemsize, _lib.CHAIN_DTYPE.fields["trans_set_value"][1]]
    assert rows[4][0] == 144                       # mmc_chain: 18 eight-byte fields


def test_header_cites_reference_lines():
    src = open(HEADER).read()
    for cite in ("Ewald/energy.jl:209-290", "Ewald/ewalds.jl:293-376", "Ewald/ewalds.jl:538-604",
                 "Ewald/ewalds.jl:718-826", "Ewald/ewalds.jl:45-103", "Ewald/ewalds.jl:829-833",
                 "Ewald/ewalds.jl:892-910", "Ewald/energy.jl:946-1032", "Ewald/energy.jl:864-943",
                 "Ewald/main.jl:621", "Ewald/main.jl:628"):
        assert cite in src, cite


def _has_gpu():
    n = C.c_int32(-1)
    st = _lib.lib().mmc_device_count(C.byref(n))
    return st == 0 and n.value > 0


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device behaviour")
def test_fails_loudly_without_a_device():
    from metropolismontecarlo_amd import api, structs
    from metropolismontecarlo_amd.device import Batch, Context
    with pytest.raises(_lib.MMCError, match="MMC_ERR_HIP"):
        Context()
    import common
    a = common.nist_arrays(1)
    with pytest.raises(_lib.MMCError, match="MMC_ERR_HIP"):
        Batch(2, a["com"], a["coords"], a["atype"], a["charge"], a["eps"], a["sig"], a["box"],
              0.28, structs.factor, 10.0, 10.0)
    moa = structs.make_moa(a["com"], a["first_atom"], a["last_atom"])
    soa = structs.make_soa(a["coords"], a["atype"], a["charge"])
    tab = structs.Tables([78.0, 0.0], [3.1, 0.0])
    with pytest.raises(_lib.MMCError):
        api.LJ_poly_ΔU(1, moa, soa, tab, 10.0, a["box"])
    ew = structs.EWALD(0.28, 5, 27, 1, [[1, 1, 1]], [0.0], [0j], [0j], structs.factor)
    with pytest.raises(_lib.MMCError):
        api.PrepareEwaldVariables(ew, a["box"])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "metropolismontecarlo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h", ".jl")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "mmc_oracle" not in txt and "from oracle" not in txt and \
                    "import oracle" not in txt, os.path.join(dirpath, f)
