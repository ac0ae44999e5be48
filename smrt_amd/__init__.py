"""smrt_amd: MI355X-native DORT hot path behind SMRT's make_model()/Model.run()/Result plugin surface."""
