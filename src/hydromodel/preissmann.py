from flowsim_amd.hydromodel.preissmann import *  # noqa: F401,F403
