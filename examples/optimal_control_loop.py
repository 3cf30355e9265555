"""Training loop shared by the two optimal-control examples (state_preparation.py, gate_optimization.py).

Adam + cosine-annealed learning rate; when the loss sits on a plateau above 0.1 the learning rate is put back to its initial value
(a warm restart), which is what lets these landscapes escape their early local minima.  The best parameter set seen is returned."""
import torch


def train(model, loss_of, epochs, lr, t_max=50, stop_below=1e-4, plateau_window=6, plateau_change=0.01, clamp=False, log_every=50):
    from pulser_diff_amd.utils import freeze_gc

    freeze_gc()  # an epoch takes milliseconds here: keep CPython's generation-2 passes (~35 ms over torch's object graph) out of the loop
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=t_max)
    history, best = [], (float("inf"), None, -1)
    for epoch in range(epochs):
        loss = loss_of(model)
        loss.backward()
        opt.step()
        opt.zero_grad()
        value = float(loss.detach())
        history.append(value)
        if value < best[0]:
            best = (value, {n: p.detach().clone() for n, p in model.named_parameters()}, epoch)
        recent = history[-(plateau_window + 1):]
        on_plateau = (len(recent) == plateau_window + 1 and value > 0.1
                      and all(abs(a - b) < plateau_change for a, b in zip(recent[1:], recent[:-1])))
        if on_plateau:
            for group in opt.param_groups:
                group["lr"] = lr
            sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=t_max)
        else:
            sched.step()
        if clamp:
            model.check_constraints()
        if value < stop_below:
            break
        model.update_sequence()
        if log_every and epoch % log_every == 0:
            print(f"  epoch {epoch:4d}  loss {value:.6f}  lr {sched.get_last_lr()[0]:.4f}")
    # NOTE: the parameters recorded at epoch e are those AFTER that epoch's step; `best[0]` is the loss measured before it.
    return best, history


def load_parameters(model, params):
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(params[n])
    model.update_sequence()
