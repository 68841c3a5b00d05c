from flowsim_amd.hydromodel.hydrograph import *  # noqa: F401,F403
