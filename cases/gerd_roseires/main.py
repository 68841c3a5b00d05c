from .model import run

if __name__ == "__main__":
    print("Start.")
    s = run(verbose=1, inflow_hyd_func=None)
    print("Finished regulated scenario; peak outflow %.1f m3/s" % s.flow[:, -1].max())
    s = run(verbose=1, inflow_hyd_func=None, with_gerd=False)
    print("Finished natural scenario; peak outflow %.1f m3/s" % s.flow[:, -1].max())
