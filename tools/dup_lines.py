"""Share of the code lines of ofx_fused25.hip (comments, blank and brace-only lines dropped) that also occur in
ofx_fused.hip, and the matching blocks of >= 5 lines (what is worth moving into ofx_fused_parts.h /
ofx_fused_*.inc).  usage: python tools/dup_lines.py"""
import difflib, re
def lines(p):
    out = []
    for l in open(p):
        t = l.split('//')[0].strip()
        if t and t not in ('{', '}', '};', '} else {', '#pragma unroll', '#pragma unroll 1', '#endif', '#else') \
                and not t.startswith(('*', '/*')):
            out.append(t)
    return out
a = lines('detprocess_amd/csrc/ofx_fused.hip'); b = lines('detprocess_amd/csrc/ofx_fused25.hip')
sa = set(a)
common = [l for l in b if l in sa]
print(f'ofx_fused.hip {len(a)} code lines, ofx_fused25.hip {len(b)}; lines of ofx_fused25.hip found in ofx_fused.hip: '
      f'{len(common)} = {100.0 * len(common) / len(b):.1f} %')
la = [l for l in b if len(l) > 25]
print(f'... of those longer than 25 characters: {len([l for l in la if l in sa])} of {len(la)} = '
      f'{100.0 * len([l for l in la if l in sa]) / len(la):.1f} %')
sm = difflib.SequenceMatcher(None, a, b, autojunk=False)
blocks = [m for m in sm.get_matching_blocks() if m.size >= 5]
print('matching blocks of >= 5 lines:', len(blocks), 'with', sum(m.size for m in blocks), 'lines')
for m in sorted(blocks, key=lambda m: -m.size)[:25]:
    print(f'  {m.size:3d} lines  fused@{m.a:5d} fused25@{m.b:5d}  {a[m.a][:80]}')
