"""cProfile of the Python thread over tools/bench_host_env.py's rollouts (where the host time of a host-env step goes).
usage: python tools/profile_host_env.py [workers] [iterations] [device_frame_stack] [groups]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_host_env

if __name__ == "__main__":
    pr = cProfile.Profile()
    pr.enable()
    bench_host_env.main()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(28)
