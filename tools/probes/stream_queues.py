"""Which HIP streams share a hardware queue?  The runtime multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues
and two streams on one queue run strictly in order, so the step's side streams only overlap if they sit on different
queues.  Probe: a long spin kernel on stream A, then an event on stream B; B's event completing while A still spins means
different queues.  Prints the equivalence classes of: the null stream, torch pool streams, freshly created HIP streams."""
import ctypes, os, sys, time
import torch

hip = ctypes.CDLL("libamdhip64.so")
N_TORCH = int(os.environ.get("N_TORCH", "8"))
N_RAW = int(os.environ.get("N_RAW", "12"))


def raw_stream(flags=1):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), flags) == 0
    return torch.cuda.ExternalStream(s.value)


def same_queue(a, b, spin=40_000_000):
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin)
        ea.record()
    with torch.cuda.stream(b):
        eb.record()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.004:
        if eb.query():
            break
    overl = eb.query() and not ea.query()
    torch.cuda.synchronize()
    return not overl


torch.cuda.init()
x = torch.zeros(1, device="cuda")
streams = [("null", torch.cuda.default_stream())]
order = os.environ.get("ORDER", "torch,raw")
for kind in order.split(","):
    if kind == "torch":
        streams += [(f"torch{i}", torch.cuda.Stream()) for i in range(N_TORCH)]
    else:
        streams += [(f"raw{i}", raw_stream()) for i in range(N_RAW)]
classes = []
for name, s in streams:
    for c in classes:
        if same_queue(c[0][1], s):
            c.append((name, s))
            break
    else:
        classes.append([(name, s)])
for i, c in enumerate(classes):
    print(f"queue class {i}: " + " ".join(n for n, _ in c))
