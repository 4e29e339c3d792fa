"""Instruction mix and resource usage of one kernel from a hipcc -S dump:  isa_stats.py file.s <mangled-name-substring> ..."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
KEYS = ['s_waitcnt', 's_load_dwordx16', 's_load_dwordx8', 's_load_dwordx4', 's_load_dwordx2', 's_load_dword', 'global_load_dwordx4',
        'global_load_dwordx2', 'global_load_dword', 'ds_read_b128', 'ds_read_b64', 'ds_read2_b64', 'ds_read_b32', 'ds_write_b128', 'ds_write_b64',
        'v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32', 'v_fma_f32', 'v_mul_f32_e32', 'v_add_f32_e32', 'v_sub_f32_e32', 'v_mov_b32_e32',
        'v_readlane_b32', 'v_writelane_b32', 'scratch_load_dword', 'scratch_store_dword', 's_barrier', 'v_mfma_f32_32x32x2_f32',
        'v_accvgpr_write_b32', 'v_accvgpr_read_b32', 's_cbranch_scc0', 's_cbranch_scc1', 's_cbranch_execz']
for name in sys.argv[2:]:
    for m in re.finditer(r'^(\S*%s\S*):[^\n]*\n' % re.escape(name), s, re.M):
        sym = m.group(1)
        end = s.find('s_endpgm', m.end())
        body = s[m.end():end]
        c = Counter(l.split()[0] for l in body.split('\n')
                    if l.strip() and not l.strip().startswith((';', '.')) and not l.strip().endswith(':'))
        print(sym, 'instructions', sum(c.values()))
        print('   ' + '  '.join('%s=%d' % (k, c[k]) for k in KEYS if c.get(k)))
        i = s.find('.amdhsa_kernel ' + sym)
        blk = s[i:i + 4000]
        out = []
        for key in ['next_free_vgpr', 'next_free_sgpr', 'private_segment_fixed_size', 'group_segment_fixed_size', 'accum_offset']:
            mm = re.search(key + r'\s+(\S+)', blk)
            out.append('%s=%s' % (key, mm.group(1) if mm else None))
        print('   ' + '  '.join(out))
