from flowsim_amd.hydromodel.utility import *  # noqa: F401,F403
