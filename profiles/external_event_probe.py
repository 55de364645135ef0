"""Where does an external event-record node of a captured hipGraph fire, seen from a stream outside the graph?  Several graph
shapes (one branch, two branches with the node on either), the stamp of a stream that waits for the event after the launch
relative to the graph's span.   python profiles/external_event_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coma_unet_amd.data_parallel import ExternalEvent  # noqa: E402

dev = torch.device("cuda")
MS = 2_000_000          # ~1 ms of torch.cuda._sleep at ~2 GHz
x = torch.zeros(1 << 20, device=dev)
y = torch.zeros(1 << 20, device=dev)
side = torch.cuda.Stream()
aux = torch.cuda.Stream()


def fork(frm, to):
    e = torch.cuda.Event()
    e.record(frm)
    to.wait_event(e)


def shape_linear(ev):
    torch.cuda._sleep(3 * MS); x.add_(1); ev.record(); torch.cuda._sleep(6 * MS); x.add_(1)


def shape_side_short(ev):       # node on a side branch that ends early
    main = torch.cuda.current_stream()
    torch.cuda._sleep(3 * MS); x.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        y.add_(1); ev.record(side); y.add_(1)
    torch.cuda._sleep(6 * MS); x.add_(1)
    fork(side, main)


def shape_side_long(ev):        # node on a side branch that runs to the end of the graph
    main = torch.cuda.current_stream()
    torch.cuda._sleep(3 * MS); x.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        y.add_(1); ev.record(side); torch.cuda._sleep(6 * MS); y.add_(1)
    torch.cuda._sleep(2 * MS); x.add_(1)
    fork(side, main)


def shape_main_two_branches(ev):  # node on the capture stream of a two-branch graph
    main = torch.cuda.current_stream()
    torch.cuda._sleep(3 * MS); x.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        torch.cuda._sleep(6 * MS); y.add_(1)
    x.add_(1); ev.record(main); torch.cuda._sleep(4 * MS); x.add_(1)
    fork(side, main)


def shape_many_kernels(ev):     # node on a side branch between many small kernels (the step's shape)
    main = torch.cuda.current_stream()
    for _ in range(50):
        x.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        for _ in range(20):
            y.add_(1)
        ev.record(side)
        for _ in range(400):
            y.add_(1)
    for _ in range(400):
        x.add_(1)
    fork(side, main)


big = torch.zeros(64 << 20, device=dev)
big2 = torch.zeros(64 << 20, device=dev)
evs = torch.cuda.Stream()


def shape_two_busy_branches_on_side(ev):    # ~100-us kernels on both branches, node between the side branch's kernels
    main = torch.cuda.current_stream()
    big.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        for _ in range(10):
            big2.add_(1)
        ev.record(side)
        for _ in range(30):
            big2.add_(1)
    for _ in range(40):
        big.add_(1)
    fork(side, main)


def shape_two_busy_branches_own_branch(ev):  # the same, node on a third branch that waits for both and carries nothing else
    main = torch.cuda.current_stream()
    big.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        for _ in range(10):
            big2.add_(1)
        fork(side, evs)
        for _ in range(30):
            big2.add_(1)
    for _ in range(10):
        big.add_(1)
    fork(main, evs)
    ev.record(evs)
    for _ in range(30):
        big.add_(1)
    fork(side, main)
    fork(evs, main)


def shape_unequal_one_node(ev):     # a long main chain, a short side chain, ONE node that depends on both
    main = torch.cuda.current_stream()
    big.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        for _ in range(3):
            big2.add_(1)
        fork(side, evs)
        for _ in range(12):
            big2.add_(1)
    for _ in range(30):
        x.add_(1)
    fork(main, evs)
    ev.record(evs)
    for _ in range(40):
        big.add_(1)
    fork(side, main)
    fork(evs, main)


def shape_unequal_on_side(ev):      # the same chains, the node in the side chain's own order
    main = torch.cuda.current_stream()
    big.add_(1)
    fork(main, side)
    with torch.cuda.stream(side):
        for _ in range(3):
            big2.add_(1)
        ev.record(side)
        for _ in range(12):
            big2.add_(1)
    for _ in range(30):
        x.add_(1)
    for _ in range(40):
        big.add_(1)
    fork(side, main)


for name, fn in [("unequal chains, one node", shape_unequal_one_node), ("unequal chains, on side", shape_unequal_on_side), ("busy branches, on side", shape_two_busy_branches_on_side), ("busy branches, own branch", shape_two_busy_branches_own_branch),("linear", shape_linear), ("side branch, short", shape_side_short), ("side branch, long", shape_side_long),
                 ("main of two branches", shape_main_two_branches), ("many kernels", shape_many_kernels)]:
    ev = ExternalEvent()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        fn(ev)
    out = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); te = torch.cuda.Event(enable_timing=True)
        t0.record()
        g.replay()
        t1.record()
        ev.wait(aux)
        te.record(aux)
        torch.cuda.synchronize()
        out.append((round(t0.elapsed_time(te), 2), round(t0.elapsed_time(t1), 2)))
    print(f"{name:24s} event seen at / graph end (ms): {out}")
