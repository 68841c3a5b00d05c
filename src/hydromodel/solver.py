from flowsim_amd.hydromodel.solver import *  # noqa: F401,F403
