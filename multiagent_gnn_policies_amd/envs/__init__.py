from .flocking import (FlockParams, VecFlock, FlockingRelativeEnv, FlockingLeaderEnv, FlockingTwoFlocksEnv,  # noqa: F401
                       FlockingStochasticEnv, TimeLimit, make, registered_ids)
