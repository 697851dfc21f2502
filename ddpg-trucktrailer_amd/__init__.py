"""MI355X-native batched truck-trailer backing environment + DDPG learner.

Drop-in for the hot path of pain7576/ddpg-trucktrailer: `Truck_trailer_Env_2` (gym-style, one
env) and `TruckTrailerVecEnv` (N envs, torch device tensors) sit on the C ABI of libttenv.so
(include/ttenv.h); `Agent` mirrors DDPG/DDPG_agent.py on PyTorch-ROCm."""
__all__ = ["TruckTrailerVecEnv", "Truck_trailer_Env_2", "Truck_trailer_Env_1"]


def __getattr__(name):
    if name == "TruckTrailerVecEnv":
        from ddpg_trucktrailer_amd.vec_env import TruckTrailerVecEnv
        return TruckTrailerVecEnv
    if name in ("Truck_trailer_Env_2", "Truck_trailer_Env_1"):
        from ddpg_trucktrailer_amd import env
        return getattr(env, name)
    raise AttributeError(name)
