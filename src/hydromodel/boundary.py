from flowsim_amd.hydromodel.boundary import *  # noqa: F401,F403
