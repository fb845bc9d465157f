"""Which hardware queue does each torch stream land on?  (dev tool: run under rocprofv3 --kernel-trace and read Queue_Id per kernel size)"""
import torch
dev = torch.device("cuda:0")
x = [torch.zeros(1 << 20, device=dev) for _ in range(12)]
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(6)] + [torch.cuda.Stream(device=dev, priority=-1) for _ in range(3)]
torch.cuda.synchronize()
for rep in range(2):
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            x[i][: (i + 1) * 1000].add_(1.0)  # kernel i is recognisable by its grid size
torch.cuda.synchronize()
print("ok")
