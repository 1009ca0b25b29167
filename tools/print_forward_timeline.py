#!/usr/bin/env python3
"""Kernel timeline of ONE model forward out of a rocprofv3 --kernel-trace CSV (tools/trace_models.sh): start offset, duration, queue, workgroups, kernel.
    python tools/print_forward_timeline.py gpurun_out/pc_c4/m_kernel_trace.csv [forward index counted from the end, default 3]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
emb = [i for i, r in enumerate(rows) if 'embedding_fwd' in r['Kernel_Name']]
# a forward starts at an embedding kernel whose predecessor embedding is more than GAP launches back (the embeddings of one forward -- question, answer, the second model's --
# sit within ~20 launches of each other; round 6: a c3 forward is ~45 launches, so the gap from its last embedding to the next forward's first is ~28)
GAP = 24
starts = [e for j, e in enumerate(emb) if j == 0 or e - emb[j - 1] > GAP]
a, b = starts[-back - 1], starts[-back]
t0 = int(rows[a]['Start_Timestamp'])


def short(n):
    n = re.sub(r'cti::\(anonymous namespace\)::', '', n)
    return re.sub(r'^void ', '', n)[:72]


busy = 0.0
for r in rows[a:b]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy += d
    print("%8.1f %7.1f q%s g%-6d %s" % (s, d, r['Queue_Id'], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), short(r['Kernel_Name'])))
print("launches %d, summed kernel time %.1f us, next forward starts at %.1f us" % (b - a, busy, (int(rows[b]['Start_Timestamp']) - t0) / 1e3))
