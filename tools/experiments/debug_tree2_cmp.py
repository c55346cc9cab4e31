import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for step in range(3):
    print("step", step, [(k, f"{((x := b['hist'][step]['grad'][lo:hi]) - (y := a['hist'][step]['grad'][lo:hi])).norm().item() / y.norm().clamp_min(1e-30).item():.4f}") for k, (lo, hi) in zip(a["keys"], a["ranges"])])
