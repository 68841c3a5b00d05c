from flowsim_amd.hydromodel.hydraulics import *  # noqa: F401,F403
