"""Debug: cfg3 trace on the GPU vs the CPU oracle, per minibatch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import load
from oracle import ppo_oracle as O
from aur_ppo_amd.ppo import ppo
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_normal_adv_tail"
z = load("trace.npz")
hp = dict(eval(str(z[f"{name}/params"])))
init = {k[len(name) + 6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{name}/init/")}
D, A = init["actor.net.0.weight"].shape[1], init["actor.net.4.weight"].shape[0]
hp.update(gym_id="Synthetic-v0", obs_dim=D, act_dim=A, log=False, save=False)
agent = ppo(hp)
agent.policy.load_state_dict(init)
agent.seed_all(1)
cont = hp["continuous"]
net = O.make_actor_critic(D, (A,) if cont else A, hp["hidden_dim"], hp["num_layers"], cont)
net.load_state_dict(init)
opt = torch.optim.Adam(net.parameters(), lr=hp["learning_rate"], eps=1e-5)
rng = np.random.RandomState(1)
U = int(z[f"{name}/num_updates"][0])
np.set_printoptions(linewidth=200, precision=6)
for u in range(U):
    lr = (1.0 - u / U) * hp["learning_rate"]
    agent.optimizer.param_groups[0]["lr"] = lr
    opt.param_groups[0]["lr"] = lr
    buf = {k: torch.from_numpy(z[f"{name}/u{u}/{k}"]) for k in ("states", "actions", "log_probs", "rewards", "terminals", "values")}
    for k, v in buf.items():
        getattr(agent.buffer, k).copy_(v)
    no, nd = torch.from_numpy(z[f"{name}/u{u}/next_obs"]), torch.from_numpy(z[f"{name}/u{u}/next_done"])
    nv_gpu = agent.policy.value(no.cuda()).detach().cpu().numpy()
    nv_cpu = net.value(no).detach().numpy()
    print(f"u{u} next_value gpu-cpu max diff", np.abs(nv_gpu - nv_cpu).max())
    ret, adv = agent.advantages(no.cuda(), nd.cuda())
    print(f"u{u} adv diff vs golden", np.abs(adv.cpu().numpy() - z[f'{name}/u{u}/advantages']).max())
    n = agent.update(ret, adv)
    res = O.reference_update(net, opt, buf, no, nd, hp, rng)
    got = agent._scalars[:n].cpu().numpy()
    print("gpu scalars\n", got)
    print("cpu scalars\n", res["scalars"])
    print("grad norms gpu", agent._norms[:n].cpu().numpy())
    for (k, v), (_, v2) in zip(agent.policy.state_dict().items(), net.state_dict().items()):
        print(f"  param {k}: max diff {np.abs(v.cpu().numpy() - v2.numpy()).max():.3e}")
