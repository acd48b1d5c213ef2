#!/usr/bin/env python3
"""Where the one-rank data-parallel step loses time against the single-replica step: reads a rocprofv3 --kernel-trace CSV of
`T3D_FORCE_DIST=1 bench.py`, takes one replayed step in the middle of the run, and prints every gap > 5 us between consecutive
kernels on the main queue plus the RCCL kernels with their start/end relative to the step.
  python tools/dp_timeline.py <kernel_trace.csv>"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    sched = [i for i, n in enumerate(names) if 'k_schedule_step' in n]
    i0, i1 = sched[len(sched) // 2], sched[len(sched) // 2 + 1]
    step = rows[i0:i1]
    t0 = int(step[0]['Start_Timestamp'])
    print('step of %d kernels, %.1f us' % (len(step), (int(rows[i1]['Start_Timestamp']) - t0) / 1e3))
    last_end, last_name = None, None
    for r in step:
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:60]
        rccl = 'ccl' in r['Kernel_Name'].lower()
        if rccl:
            print('   RCCL  %-60s queue %s start %8.1f end %8.1f (%.1f us)' % (n, r.get('Queue_Id', '?'), s, e, e - s))
            continue
        if last_end is not None and s - last_end > 5.0:
            print('   gap %6.1f us before %-50s (after %s), at %.1f' % (s - last_end, n, last_name, s))
        last_end, last_name = e, n


if __name__ == '__main__':
    main()
