bash tools/profile_round.sh
