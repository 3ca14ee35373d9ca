import torch, sys
sys.path.insert(0,'.')
import tests.test_gpu_ops as T
# monkeypatch asserts: rerun body with prints
import inspect, re
src = inspect.getsource(T.test_sbf_project_twice_differentiable_matches_float64)
src = src.split('\n',1)[1]  # drop decorator
src = src.replace("assert (a.detach().cpu().double() - r.detach()).abs().max() <= 5e-6 * r.abs().max().clamp(min=1.0)", "print('P', ((a.detach().cpu().double() - r.detach()).abs().max()/r.abs().max()).item())")
src = src.replace("assert (fb.detach().cpu().double() - fb64.detach()).abs().max() <= 1e-5 * fb64.abs().max()", "print('fb', ((fb.detach().cpu().double() - fb64.detach()).abs().max()/fb64.abs().max()).item())")
src = src.replace("assert (fa.detach().cpu().double() - fa64.detach()).abs().max() <= 1e-5 * fa64.abs().max()", "print('fa', ((fa.detach().cpu().double() - fa64.detach()).abs().max()/fa64.abs().max()).item())")
src = src.replace("assert (a.grad.cpu().double() - r.grad).abs().max() <= 2e-5 * r.grad.abs().max().clamp(min=1.0), (name, deferred)", "print(name, deferred, ((a.grad.cpu().double() - r.grad).abs().max()/r.grad.abs().max()).item())")
ns = {}
exec(src, T.__dict__, ns)
ns['test_sbf_project_twice_differentiable_matches_float64'](7,6,4,'md17_b8')
