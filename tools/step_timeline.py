#!/usr/bin/env python3
"""One replayed step of a rocprofv3 --kernel-trace CSV as a timeline: every kernel with its queue, start offset and duration, the idle
time of each queue, and what ran on the other queue(s) meanwhile.  The step is the one between two `k_schedule_step` launches in the
middle of the run.
  python tools/step_timeline.py <kernel_trace.csv> [--which k]"""
import argparse
import csv


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:58]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--which', type=int, default=None, help='index of the step (default: the middle one)')
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    sched = [i for i, r in enumerate(rows) if 'k_schedule_step' in r['Kernel_Name']]
    k = a.which if a.which is not None else len(sched) // 2
    i0, i1 = sched[k], sched[k + 1]
    step = rows[i0:i1]
    t0 = int(step[0]['Start_Timestamp'])
    length = (int(rows[i1]['Start_Timestamp']) - t0) / 1e3
    queues = sorted({r.get('Queue_Id', '?') for r in step})
    print('step %d: %d kernels, %.1f us, queues %s' % (k, len(step), length, queues))
    busy = {q: 0.0 for q in queues}
    last_end = {q: None for q in queues}
    for r in step:
        q = r.get('Queue_Id', '?')
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        gap = '' if last_end[q] is None else ('gap %5.1f' % (s - last_end[q]))
        others = [short(o['Kernel_Name'])[:24] for o in step if o is not r and o.get('Queue_Id', '?') != q
                  and int(o['Start_Timestamp']) < int(r['End_Timestamp']) and int(o['End_Timestamp']) > int(r['Start_Timestamp'])]
        print('%s%-3s %8.1f +%7.1f  %-9s %-58s %s' % ('    ' * queues.index(q), q, s, e - s, gap, short(r['Kernel_Name']),
                                                     ('|| ' + ', '.join(others[:3])) if others else ''))
        busy[q] += e - s
        last_end[q] = e
    for q in queues:
        print('queue %s busy %.1f us of %.1f' % (q, busy[q], length))


if __name__ == '__main__':
    main()
