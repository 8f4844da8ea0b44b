from .a2c_acktr import A2C_ACKTR
from .ppo import PPO
