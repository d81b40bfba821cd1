"""Guard rails around the compiler.  Round 3's wrong window-attention gradients came from hipcc
re-using the index register of a `ds_bpermute_b32` still in flight for a `v_accvgpr_read_b32`; `tools/isa_lint.py` finds
that pattern in compiler output.  `__graft_entry__.build()` keeps the device assembly of every translation unit
(-save-temps=obj) and fails on a finding; these tests run the same scan over the built tree, check that the scanner
still recognises the pattern, and that no kernel in the library spills registers to scratch."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import __graft_entry__ as entry  # noqa: E402
import isa_lint  # noqa: E402


def test_every_translation_unit_is_linted_and_clean():
    entry.build()                       # no-op when the tree is built; raises on a finding
    srcs = glob.glob(os.path.join(ROOT, 'torchok_amd', 'csrc', '*.hip'))
    assert len(srcs) >= 15
    objdir = os.path.join(ROOT, 'torchok_amd', 'lib', 'obj')
    for s in srcs:
        asm = entry._asm_of(os.path.join(objdir, os.path.basename(s) + '.o'))
        assert os.path.exists(asm), asm
        assert os.path.getmtime(asm) >= os.path.getmtime(s), f'{asm} is older than its source'
    sites, spills = entry.lint_isa(raise_on_findings=False)
    assert sites == []
    assert spills == {}


def test_lint_recognises_the_round3_pattern(tmp_path):
    bad = tmp_path / 'bad.s'
    bad.write_text('\n'.join([
        '_Z4kernv:',
        '\tds_bpermute_b32 v152, v20, v144',
        '\tds_bpermute_b32 v153, v20, v145',
        '\tv_accvgpr_read_b32 v20, a6',
        '\ts_waitcnt lgkmcnt(0)',
        '\tds_bpermute_b32 v1, v2, v3',
        '\ts_waitcnt lgkmcnt(0)',
        '\tv_accvgpr_read_b32 v2, a7',      # behind the wait: fine
        '\ts_endpgm', '']))
    sites = isa_lint.lint_file(str(bad))
    assert len(sites) == 2 and all(s[2] == '_Z4kernv' for s in sites)       # both permutes read v20
    assert all('v_accvgpr_read_b32 v20' in s[4] for s in sites)


def test_kernel_resources_reads_spill_counts(tmp_path):
    f = tmp_path / 'k.s'
    f.write_text('\n'.join([
        '\t.amdhsa_kernel _Z1kv', '; Kernel info:', '; NumVgprs: 256', '; NumAgprs: 0', '; ScratchSize: 240', '; Occupancy: 2',
        'amdhsa.kernels:', '    .name:           _Z1kv', '    .sgpr_spill_count: 0', '    .vgpr_spill_count: 58', '']))
    r = isa_lint.kernel_resources(str(f))['_Z1kv']
    assert r['spill_vgpr'] == 58 and r['scratch'] == 240 and r['vgpr'] == 256
