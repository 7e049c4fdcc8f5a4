"""CPU tests (-m "not gpu") of the drop-in boundary: the C-ABI library loads without a GPU,
exports every symbol include/*.h declares, and fails LOUDLY (no CPU fallback) when asked to
compute without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in ("srx.h", "srx_synth.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(srx_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from singlerust_amd import build, _ffi
    build.build()
    return _ffi.lib()


def test_library_exports_every_declared_symbol(lib):
    from singlerust_amd import _ffi
    decl = declared_functions()
    assert len(decl) >= 35
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    # and the ctypes binding covers the whole header
    assert set(decl) == set(_ffi.EXPORTS)


def test_rust_bindings_follow_the_header():
    """rust/srx_sys.rs (the `extern "C"` module of INTEGRATION.md) is generated from include/srx.h: the committed
    file must be what the generator produces now, and declare every function of that header with its arity."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(ROOT, "scripts", "gen_rust_bindings.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = open(os.path.join(ROOT, "rust", "srx_sys.rs")).read()
    assert text == gen.render(), "rust/srx_sys.rs is stale: run python scripts/gen_rust_bindings.py"
    header = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "srx.h")).read(), flags=re.S)
    in_header = {m.group(1): m.group(2) for m in re.finditer(r"\b(srx_\w+)\s*\(([^()]*)\)\s*;", header)}
    in_header = {k: v for k, v in in_header.items() if not k.endswith("_fn")}
    in_rust = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (srx_\w+)\((.*?)\)(?: -> [^;]+)?;", text, flags=re.S)}
    assert set(in_rust) == set(in_header) and len(in_rust) >= 50

    def arity(params, sep_depth_char="("):
        params = params.strip()
        if not params or params == "void":
            return 0
        depth, n = 0, 1
        for ch in params:
            depth += ch == "("
            depth -= ch == ")"
            n += ch == "," and depth == 0
        return n
    for name in in_header:
        assert arity(in_header[name]) == arity(in_rust[name]), name
    # the hand-written shim only calls functions the generated module declares
    shim = open(os.path.join(ROOT, "rust", "shim.rs")).read()
    assert set(re.findall(r"\b(srx_\w+)\(", shim)) <= set(in_rust)


def test_abi_version(lib):
    assert lib.srx_abi_version() == 6


def test_struct_layouts_match_header():
    """sizeof() of the ctypes mirrors equals what the C compiler lays out (checked with gcc)."""
    import subprocess, tempfile
    from singlerust_amd import _ffi
    prog = r'''
#include <stdio.h>
#include "srx.h"
#include "srx_synth.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu\n", sizeof(srx_csr), sizeof(srx_mat_info), sizeof(srx_pca_opts),
 sizeof(srx_pca_info), sizeof(srx_pipeline_result), sizeof(srx_synth_params)); return 0;}'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "t")]).split()]
    mirrors = [_ffi.Csr, _ffi.MatInfo, _ffi.PcaOpts, _ffi.PcaInfo, _ffi.PipelineResult, _ffi.SynthParams]
    assert sizes == [C.sizeof(m) for m in mirrors]


def test_partition_rows_is_nnz_balanced(lib):
    from singlerust_amd import _ffi
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 200, 5000)
    lens[100:200] = 3000                                     # a heavy stripe
    indptr = np.zeros(5001, dtype=np.uint64)
    indptr[1:] = np.cumsum(lens)
    for n in (1, 2, 3, 8):
        cut = np.zeros(n + 1, dtype=np.uint64)
        assert lib.srx_partition_rows(_ffi.ptr(indptr), 5000, n, _ffi.ptr(cut)) == 0
        assert cut[0] == 0 and cut[-1] == 5000 and np.all(np.diff(cut.astype(np.int64)) >= 0)
        per = np.diff(indptr[cut.astype(np.int64)].astype(np.int64))
        assert per.sum() == indptr[-1]
        assert per.max() - per.min() <= 2 * 3000             # within one heavy row of balanced


def test_synth_host_generator_is_canonical_csr(lib):
    from singlerust_amd import _ffi
    p = _ffi.SynthParams()
    lib.srx_synth_defaults(C.byref(p), 11, 3000, 4000, 0.05)
    indptr = np.zeros(3001, dtype=np.uint64)
    assert lib.srx_synth_indptr(C.byref(p), 0, 3000, _ffi.ptr(indptr)) == 0
    nnz = int(indptr[-1])
    idx = np.zeros(nnz, dtype=np.uint64)
    val = np.zeros(nnz, dtype=np.float32)
    assert lib.srx_synth_fill_host(C.byref(p), 0, 3000, _ffi.ptr(indptr), _ffi.ptr(idx), _ffi.ptr(val)) == 0
    assert idx.max() < 4000 and val.min() >= 1
    lens = np.diff(indptr.astype(np.int64))
    assert abs(lens.mean() / 4000 - 0.05) < 0.01
    for i in range(3000):
        assert np.all(np.diff(idx[indptr[i]:indptr[i + 1]].astype(np.int64)) > 0)   # sorted + unique
    # any row range is reproducible on its own (shard-local generation)
    ip2 = np.zeros(501, dtype=np.uint64)
    lib.srx_synth_indptr(C.byref(p), 1000, 1500, _ffi.ptr(ip2))
    idx2 = np.zeros(int(ip2[-1]), dtype=np.uint64)
    val2 = np.zeros(int(ip2[-1]), dtype=np.float32)
    lib.srx_synth_fill_host(C.byref(p), 1000, 1500, _ffi.ptr(ip2), _ffi.ptr(idx2), _ffi.ptr(val2))
    a, b = int(indptr[1000]), int(indptr[1500])
    assert np.array_equal(idx2, idx[a:b]) and np.array_equal(val2, val[a:b])


def test_no_cpu_fallback_without_gpu(lib):
    """Without a device every compute path must raise — never silently compute on the host."""
    from singlerust_amd import _ffi
    n = C.c_int32(0)
    rc = lib.srx_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    import singlerust_amd as sr
    with pytest.raises(sr.SrxError) as ei:
        sr.Context(0)
    assert ei.value.code == _ffi.E_HIP and "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    """The product package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "singlerust_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower().replace("no cpu oracle", ""), f"{f} mentions the oracle"


def test_integration_excerpts_are_the_shim():
    """INTEGRATION.md shows excerpts of rust/shim.rs, not a second version of it: every fenced rust block whose first
    line is `// rust/shim.rs ...` must appear verbatim (after that line) in the file."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(os.path.join(ROOT, "rust", "shim.rs")).read()
    blocks = re.findall(r"```rust\n// rust/shim\.rs[^\n]*\n(.*?)```", doc, flags=re.S)
    assert len(blocks) >= 7
    for b in blocks:
        assert b.rstrip("\n") in shim, b[:200]
    # and the free functions carry the reference's signatures (src/memory/processing/mod.rs:303-332, dim_red/mod.rs:24)
    for sig in ("pub fn normalize_total_inplace(adata: &mut IMAnnData, target_sum: f64, direction: Direction) -> anyhow::Result<()>",
                "pub fn normalize_total(adata: &IMAnnData, target_sum: f64, direction: Direction) -> anyhow::Result<IMAnnData>",
                "pub fn log1p_transform_inplace(adata: &mut IMAnnData) -> anyhow::Result<()>",
                "pub fn log1p_transform(adata: &IMAnnData) -> anyhow::Result<IMAnnData>",
                "pub fn compute_number(adata: &IMAnnData, direction: Direction) -> anyhow::Result<Vec<u32>>",
                "pub fn compute_variance(adata: &IMAnnData, direction: Direction) -> anyhow::Result<Vec<f64>>",
                "pub fn pca_inplace<S: SVDImplementation>(anndata: &mut IMAnnData, n_components: Option<usize>, center: Option<bool>,",
                # the (f) rows: src/memory/processing/mod.rs:86,120,245,273, src/memory/statistics/mod.rs:48,74,
                # src/backed/statistics/mod.rs:5,26
                "pub fn filter_cells_inplace(adata: &mut IMAnnData, lower_lim: FlexValue, upper_lim: FlexValue) -> anyhow::Result<()>",
                "pub fn filter_cells(adata: &IMAnnData, lower_lim: FlexValue, upper_lim: FlexValue) -> anyhow::Result<IMAnnData>",
                "pub fn filter_genes_inplace(adata: &mut IMAnnData, lower_lim: FlexValue, upper_lim: FlexValue) -> anyhow::Result<()>",
                "pub fn filter_genes(adata: &IMAnnData, lower_lim: FlexValue, upper_lim: FlexValue) -> anyhow::Result<IMAnnData>",
                "pub fn compute_qc_variables(adata: &IMAnnData) -> anyhow::Result<crate::memory::statistics::StatisticsContainer>",
                "pub fn qc_vars_inplace(adata: &IMAnnData) -> anyhow::Result<()>",
                "pub fn compute_number<B: Backend>(adata: AnnData<B>, direction: Direction, mode: ComputationMode) -> anyhow::Result<Vec<u32>>",
                "pub fn compute_sum<B: Backend>(adata: AnnData<B>, direction: Direction, mode: ComputationMode) -> anyhow::Result<Vec<f64>>"):
        assert sig in shim, sig


def test_hot_kernels_do_not_spill_registers(tmp_path):
    """The Gram stripe kernel runs at a 64-VGPR budget with two register sets of loads in flight: a one-line change (a 64-bit
    multiply in the load address, a store through what had been a read-only pointer) tipped its f32 instantiation into 56
    spilled VGPRs and cost 0.3 ms per step before any test noticed (round 5).  Compile the device code of the two files that
    hold the three largest kernels and read the code-object metadata: no VGPR spill, no scratch."""
    import concurrent.futures as cf
    import subprocess
    from singlerust_amd import build
    csrc = os.path.join(ROOT, "singlerust_amd", "csrc")
    wanted = {"pca_form.hip": ["k_gram_stripesIfE", "k_gram_stripesIdE", "k_rowcount_listEPKl"],
              "pca_solve.hip": ["k_spmm_rowsIffLi4ELb0ELi4EE", "k_spmm_rangesIffLi2ELi1024ELi2EE"]}

    def asm(src):
        out = str(tmp_path / (src + ".s"))
        subprocess.check_call([build._hipcc(), *[f for f in build.FLAGS if f != "-Wall"], "-w", "-S", "--cuda-device-only", "-o", out,
                               os.path.join(csrc, src)], stderr=subprocess.DEVNULL)
        return open(out).read()
    with cf.ThreadPoolExecutor(max_workers=2) as ex:
        texts = dict(zip(wanted, ex.map(asm, wanted)))
    for src, kernels in wanted.items():
        meta = texts[src]
        for kname in kernels:
            m = re.search(r"\.name:\s+(\S*%s\S*)\n(.*?)\.wavefront_size" % re.escape(kname), meta, flags=re.S)
            assert m, f"{kname} not found in {src}"
            body = m.group(2)
            spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", body).group(1))
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", body).group(1))
            assert spill == 0 and scratch == 0, f"{m.group(1)}: {spill} spilled VGPRs, {scratch} B of scratch"
    _check_count_pass_listing(texts["pca_form.hip"])
    for mangled, min_loads in (("k_gram_stripesIfE", 32), ("k_rowcount_listEPKl", 32)):
        _check_no_use_of_registers_in_flight(texts["pca_form.hip"], mangled, min_loads)
    _check_stripe_kernel_drains_its_lds_atomics(texts["pca_form.hip"])


def _check_count_pass_listing(text):
    """k_rowcount_list issues the 16 loads of a batch in one assembly statement and waits for them by a count, one batch later
    (`s_waitcnt vmcnt(16)` + two empty statements that hand the registers over).  The compiler believes the registers are valid
    from the load statement on: a copy it placed between the loads and the hand-over (phi resolution, live-range splitting)
    would read registers whose loads are still in flight — a first form of the kernel did exactly that on one path.  Walk the
    listing: between a load block and the hand-over that follows it in the text, no instruction outside the assembly blocks
    may name a register the block loads into."""
    for mangled in ("k_rowcount_listEPKl",):
        m = re.search(r"^(_ZN3srx\S*%s\S*):.*?s_endpgm" % mangled, text, flags=re.S | re.M)
        assert m, mangled
        lines = m.group(0).split("\n")
        i, blocks = 0, 0
        while i < len(lines):
            if "#ASMSTART" in lines[i] and i + 1 < len(lines) and re.search(r"global_load_(ushort|dword) v\d+, v\d+, s\[", lines[i + 1]):
                dests = set()
                j = i + 1
                while "#ASMEND" not in lines[j]:
                    dests.add(re.search(r"global_load_\w+ (v\d+),", lines[j]).group(1))
                    j += 1
                assert len(dests) == 16, (mangled, dests)
                blocks += 1
                # forward to the hand-over: an empty assembly statement
                k_, inside = j + 1, False
                while not ("#ASMSTART" in lines[k_] and "#ASMEND" in lines[k_ + 1]):
                    ln = lines[k_]
                    if "#ASMSTART" in ln:
                        inside = True
                    elif "#ASMEND" in ln:
                        inside = False
                    elif not inside and not ln.strip().startswith((";", ".")) and ln.strip():
                        regs = set(re.findall(r"\bv(\d+)\b", ln)) | {str(x) for a, b in re.findall(r"v\[(\d+):(\d+)\]", ln)
                                                                    for x in range(int(a), int(b) + 1)}
                        hit = {f"v{r}" for r in regs} & dests
                        assert not hit, f"{mangled}: `{ln.strip()}` touches {sorted(hit)} while their loads are in flight"
                    k_ += 1
                    assert k_ + 1 < len(lines), f"{mangled}: no hand-over behind a load block"
                i = j
            i += 1
        assert blocks >= 3, (mangled, blocks)       # before the loop, and one per half of its body


def _vregs(ln):
    r = set(int(x) for x in re.findall(r"\bv(\d+)\b", ln))
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", ln):
        r |= set(range(int(a), int(b) + 1))
    return r


def _check_no_use_of_registers_in_flight(text, mangled, min_asm_loads):
    """The inline-assembly cores (k_gram_stripes<float>, k_rowcount_list) issue global loads the compiler cannot see and wait for
    them with hard-coded `s_waitcnt vmcnt(N)`.  Memory operations return in order, so the listing can be replayed: a
    queue of the destination registers of every vector-memory operation in flight (assembly blocks and compiler code alike; stores
    count), popped by every `s_waitcnt vmcnt(N)` down to its N youngest.  No instruction OUTSIDE an assembly block may name a
    register that is still in the queue (a copy placed there by phi resolution or live-range splitting would read or clobber a
    register whose load has not landed).  Linear walk of the text: exact for the unrolled bodies, an approximation at loop back edges."""
    m = re.search(r"^(_ZN3srx\S*%s\S*):.*?s_endpgm" % mangled, text, flags=re.S | re.M)
    assert m, mangled
    queue, inside, n_asm_loads = [], False, 0
    for ln in m.group(0).split("\n"):
        s_ = ln.strip()
        if "#ASMSTART" in s_:
            inside = True
            continue
        if "#ASMEND" in s_:
            inside = False
            continue
        if not s_ or s_.startswith((";", ".")) or s_.endswith(":"):
            continue
        w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", s_)
        if w:
            n = int(w.group(1))
            queue = queue[len(queue) - n:] if 0 < n < len(queue) else ([] if n == 0 else queue)
            continue
        in_flight = set().union(*queue) if queue else set()
        if re.match(r"(global|buffer|scratch|flat)_(load|store|atomic)", s_):
            dst = _vregs(s_.split()[1].rstrip(",")) if re.match(r"\w+_load", s_) else set()
            if inside and dst:
                n_asm_loads += 1
            hit = (_vregs(s_) - dst) & in_flight
            assert inside or not hit, f"{mangled}: `{s_}` reads {sorted(hit)} while their loads are in flight"
            queue.append(dst)
            continue
        if inside:
            continue                   # the assembly blocks carry their own waits
        hit = _vregs(s_) & in_flight
        assert not hit, f"{mangled}: `{s_}` touches v{sorted(hit)} while their loads are in flight"
    assert n_asm_loads >= min_asm_loads, (mangled, n_asm_loads)


def _check_stripe_kernel_drains_its_lds_atomics(text):
    """The stripe kernel's ds_add_u64 sit in assembly blocks: the drain before the flush's barrier is an explicit
    `s_waitcnt lgkmcnt(0)` of its own, not whatever the compiler's fence happens to emit."""
    m = re.search(r"^(_ZN3srx\S*k_gram_stripesIfE\S*):.*?s_endpgm", text, flags=re.S | re.M)
    assert m
    lines = m.group(0).split("\n")
    assert any("ds_add_u64" in ln for ln in lines)
    found = False
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln and "s_waitcnt lgkmcnt(0)" in lines[i + 1] and "#ASMEND" in lines[i + 2]:
            found = found or any("s_barrier" in x for x in lines[i + 3:i + 12])
    assert found


def test_graft_entry_build_runs():
    """The driver's build check (`__graft_entry__.build()`: compile every HIP source, build the oracle, import the package, compare
    the library's ABI version with the header's) must pass on CPU — it once carried a literal version and would have failed the
    round's build check after an ABI bump."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    ge.build()
