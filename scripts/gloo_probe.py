"""Dev probe: gloo collectives on CUDA tensors, two ranks on one GPU (what the bench rehearsal relies on)."""
import faulthandler, os, torch, torch.distributed as dist
faulthandler.dump_traceback_later(60, exit=True)
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
aux = torch.cuda.Stream()
x = torch.full((334, 10), float(r), device="cuda"); out = torch.empty(668, 10, device="cuda")
c = torch.full((334,), r, dtype=torch.int32, device="cuda"); cout = torch.empty(668, dtype=torch.int32, device="cuda")
g = torch.ones(600000, device="cuda")
for i in range(3):
    with torch.cuda.stream(aux):
        dist.all_gather_into_tensor(out, x)
        dist.all_gather_into_tensor(cout, c)
    print(r, i, "gathered", float(out.sum()), int(cout.sum()), flush=True)
    dist.all_reduce(g)
    print(r, i, "reduced", float(g[0]), flush=True)
dist.barrier()
