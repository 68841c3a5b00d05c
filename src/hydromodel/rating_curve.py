from flowsim_amd.hydromodel.rating_curve import *  # noqa: F401,F403
