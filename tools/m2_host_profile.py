import cProfile, pstats, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["m2", "30", "8", "bf16", "1"]
import tools.m2_train_step as t
pr = cProfile.Profile()
pr.enable()
t.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
