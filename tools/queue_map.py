"""Which hardware queue does each torch stream land on?  Run under `rocprofv3 --kernel-trace --output-format csv`; prints nothing itself:
the trace's Queue_Id / Stream_Id columns are the answer (tools/_run_r3s.sh summarises them)."""
import sys
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
x.add_(1)                      # null stream first
streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
for i, s in enumerate(streams):
    with torch.cuda.stream(s):
        y = torch.full((256 + i,), float(i), device=dev)   # fill kernel with a grid size that names the stream
torch.cuda.synchronize()
for i, s in enumerate(streams):    # second round: does the mapping stay?
    with torch.cuda.stream(s):
        y = torch.full((512 + i,), float(i), device=dev)
torch.cuda.synchronize()
