from .flocking import (FlockParams, VecFlock, FlockingRelativeEnv, FlockingLeaderEnv, FlockingTwoFlocksEnv,  # noqa: F401
                       TimeLimit, make, registered_ids)
