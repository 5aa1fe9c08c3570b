import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import promonet_amd, restatement as oracle
device = torch.device('cuda:0')
golden = torch.load('tests/golden/generator_default.pt', weights_only=False)
state = oracle.random_state_fargan(seed=0)
state['pitch_distribution'] = golden['pitch_distribution'].clone()
inputs = oracle.synthetic_inputs(32, 861, seed=55)
pick = [3, 30, 11, 20]
with torch.inference_mode():
    want = oracle.fargan_generator_forward(*[t[pick] for t in inputs], state)
for dtype in ('fp32', 'mixed', 'f16'):
    promonet_amd.configure(MODEL='fargan', FARGAN_WEIGHT_DTYPE=dtype)
    model = promonet_amd.model.Generator(); model.load_state_dict(state); model = model.to(device).eval()
    with torch.inference_mode():
        got = model(*[t.to(device) for t in inputs], None).cpu()
    d = (got[pick] - want).abs()
    per = d.reshape(len(pick), 861, 256).amax(-1)
    print(dtype, 'max-abs', d.max().item(), 'rms', d.pow(2).mean().sqrt().item(), 'abs-max out', want.abs().max().item())
    print('   max-abs by 100-frame block:', [round(per[:, i:i+100].max().item(), 6) for i in range(0, 861, 100)])
