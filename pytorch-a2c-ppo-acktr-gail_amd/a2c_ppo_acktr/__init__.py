"""a2c_ppo_acktr -- MI355X-native drop-in for the learner update path of
ikostrikov/pytorch-a2c-ppo-acktr-gail (same import paths as the reference)."""
