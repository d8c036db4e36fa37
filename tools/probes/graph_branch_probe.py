"""How a replayed hipGraph schedules independent branches (GPU box): python tools/graph_branch_probe.py
Each scenario captures spin kernels (one workgroup each: they never compete for CUs) on two or three streams with the fork / join
pattern named, replays the graph and prints the replay's GPU time beside the serial sum and the critical path of the DAG."""
import torch

dev = torch.device("cuda:0")
CYC_PER_MS = None


def spin(ms):
    torch.cuda._sleep(int(ms * CYC_PER_MS))


def calibrate():
    global CYC_PER_MS
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000); torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(20000000); e1.record(); torch.cuda.synchronize()
    CYC_PER_MS = 20000000 / e0.elapsed_time(e1)


def run(name, body, serial, critical):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            body(s)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("%-64s replay %6.2f ms   serial %5.1f   critical path %5.1f" % (name, min(ts), serial, critical))


def main():
    calibrate()
    A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def s1(m):      # side branch captured BEFORE the main continuation
        spin(0.2); A.wait_stream(m)
        with torch.cuda.stream(A):
            spin(2.0)
        spin(2.0); m.wait_stream(A); spin(0.2)
    run("fork: side first, then main; join", s1, 4.4, 2.4)

    def s2(m):      # main continuation captured first
        spin(0.2); ev = torch.cuda.Event(); ev.record(m)
        spin(2.0)
        A.wait_event(ev)
        with torch.cuda.stream(A):
            spin(2.0)
        m.wait_stream(A); spin(0.2)
    run("fork: main first, then side (event); join", s2, 4.4, 2.4)

    def s3(m):      # the step's pattern: many short fork-joins (weight gradients), then a long side chain, more fork-joins, late join
        for _ in range(6):
            spin(0.1); A.wait_stream(m)
            with torch.cuda.stream(A):
                spin(0.1)
            spin(0.1); m.wait_stream(A)
        B.wait_stream(m)
        with torch.cuda.stream(B):
            for _ in range(10):
                spin(0.2)
        for _ in range(10):
            spin(0.1); A.wait_stream(m)
            with torch.cuda.stream(A):
                spin(0.1)
            spin(0.1); m.wait_stream(A)
        m.wait_stream(B); spin(0.1)
    run("6 fork-joins, long chain on a 3rd stream, 10 fork-joins, join", s3, 6 * 0.3 + 2.0 + 10 * 0.3 + 0.1, 6 * 0.2 + 10 * 0.2 + 0.1)

    def s3b(m):     # same, the long chain captured AFTER everything else (its dependency is an event recorded where it forks)
        for _ in range(6):
            spin(0.1); A.wait_stream(m)
            with torch.cuda.stream(A):
                spin(0.1)
            spin(0.1); m.wait_stream(A)
        ev = torch.cuda.Event(); ev.record(m)
        for _ in range(10):
            spin(0.1); A.wait_stream(m)
            with torch.cuda.stream(A):
                spin(0.1)
            spin(0.1); m.wait_stream(A)
        B.wait_event(ev)
        with torch.cuda.stream(B):
            for _ in range(10):
                spin(0.2)
        m.wait_stream(B); spin(0.1)
    run("same, long chain captured last (forks from an event)", s3b, 6 * 0.3 + 2.0 + 10 * 0.3 + 0.1, 6 * 0.2 + 10 * 0.2 + 0.1)

    def s4(m):      # the long chain waits on TWO main-stream events (gradients of two scales)
        spin(0.1); e1 = torch.cuda.Event(); e1.record(m)
        spin(0.1); e2 = torch.cuda.Event(); e2.record(m)
        B.wait_event(e1); B.wait_event(e2)
        with torch.cuda.stream(B):
            for _ in range(10):
                spin(0.2)
        for _ in range(10):
            spin(0.2)
        m.wait_stream(B); spin(0.1)
    run("chain waiting on two events, captured before the main chain", s4, 0.2 + 2.0 + 2.0 + 0.1, 0.2 + 2.0 + 0.1)

    def s5(m):      # three independent chains forked at once (the three scales of the head)
        spin(0.1); A.wait_stream(m); B.wait_stream(m)
        with torch.cuda.stream(A):
            for _ in range(5):
                spin(0.2)
        with torch.cuda.stream(B):
            for _ in range(5):
                spin(0.2)
        for _ in range(5):
            spin(0.2)
        m.wait_stream(A); m.wait_stream(B); spin(0.1)
    run("three chains forked at once; join", s5, 0.1 + 3.0 + 0.1, 0.1 + 1.0 + 0.1)

    def s6(m):      # nested: inside the side chain, short fork-joins to the weight-gradient stream
        spin(0.1); B.wait_stream(m)
        with torch.cuda.stream(B):
            for _ in range(5):
                spin(0.1); A.wait_stream(B)
                with torch.cuda.stream(A):
                    spin(0.1)
                spin(0.1); B.wait_stream(A)
        for _ in range(5):
            spin(0.3)
        m.wait_stream(B); spin(0.1)
    run("side chain with nested fork-joins beside a main chain", s6, 0.1 + 1.5 + 1.5 + 0.1, 0.1 + 1.5 + 0.1)


if __name__ == "__main__":
    main()
