from flowsim_amd.hydromodel.lumped_storage import *  # noqa: F401,F403
