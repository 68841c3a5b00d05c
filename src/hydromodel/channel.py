from flowsim_amd.hydromodel.channel import *  # noqa: F401,F403
