from flowsim_amd.hydromodel.cross_section import *  # noqa: F401,F403
