"""Host profile of steady-state stage-2 training steps (tools; GPU box): python tools/m2_host_profile.py [philox|cpu|device]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("M2_DROPOUT", sys.argv[1] if len(sys.argv) > 1 else "philox")
import tools.m2_train_step as t  # noqa: E402
sys.argv = ["m2", "10", "8", "bf16", "1"]
t.main()                      # warm: plans, autotuning, first-call costs
pr = cProfile.Profile()
sys.argv = ["m2", "30", "8", "bf16", "1"]
pr.enable()
t.main()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
